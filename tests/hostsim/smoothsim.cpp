// TEST INFRASTRUCTURE ONLY (never part of the product library, never a fallback).
// The dense-powers one-launch smoother (tgp_modal.hip k_smooth_one + tgp_steady_plan.hpp build_smooth / smooth_head_*) run on the
// host: the product's own plan and head functions, and a restatement of the kernel's orchestration -- spans with halos at both ends, a lane's
// eight steps from a zero state, the in-row and across-row scan levels with the very tables the kernel uses (P[k], the per-lane table built
// from the bits of the lane number, PT), the tiles chained over the three before / behind them, the WJ / WG rows -- lane by lane.
// What is NOT exercised here is the kernel's own code; the GPU tier covers it.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../temporalgps.jl_amd/csrc/tgp_steady_plan.hpp"

using namespace tgp_plan;

namespace {

template <int D>
struct Tables {
    double pw[64][D][D], pg[64][D][D];      // Phi^(8 e), G^(8 e) by the kernel's bit method
};

template <int D>
void lane_table(const double (*P)[kRandMaxD * kRandMaxD], double (&out)[64][D][D]) {
    for (int lane = 0; lane < 64; ++lane)
        for (int i = 0; i < D; ++i) {
            double row[D];
            for (int k = 0; k < D; ++k) row[k] = (k == i) ? 1.0 : 0.0;
            for (int b = 0; b < 6; ++b) {
                double nr[D];
                for (int k = 0; k < D; ++k) {
                    double v = 0.0;
                    for (int m = 0; m < D; ++m) v = std::fma(row[m], P[b][m * D + k], v);
                    nr[k] = v;
                }
                if ((lane >> b) & 1)
                    for (int k = 0; k < D; ++k) row[k] = nr[k];
            }
            for (int k = 0; k < D; ++k) out[lane][i][k] = row[k];
        }
}

template <int D>
void matvec_acc(const double* M /*row-major D x D*/, const double* x, double* y) {      // y += M x
    for (int i = 0; i < D; ++i) {
        double v = y[i];
        for (int k = 0; k < D; ++k) v = std::fma(M[i * D + k], x[k], v);
        y[i] = v;
    }
}
template <int D>
void matvec_acc2(const double (&M)[D][D], const double* x, double* y) {
    for (int i = 0; i < D; ++i) {
        double v = y[i];
        for (int k = 0; k < D; ++k) v = std::fma(M[i][k], x[k], v);
        y[i] = v;
    }
}

template <int D>
int run_d(const ModelHost& m, long long T, const double* y, const double* Rnew, int rnew_per_step, double* mean, double* var, double* out,
          const double* eps_t = nullptr, const double* eps_e = nullptr, const double* eps_0 = nullptr, const double* hh_t = nullptr) {      // hh_t: an emission offset per step
    const bool rnd = eps_t != nullptr;      // a draw from the posterior (k_smooth_one<..., RAND>): `mean` receives it
    static thread_local double tvb[kTailMax];
    SmoothPlan sp;
    build_smooth<D>(m, T, sp, tvb);
    out[1] = sp.why;
    if (sp.why != kOk) return 0;
    const FilterPlan& fp = sp.fp;
    out[2] = fp.n0; out[3] = fp.nhs; out[4] = sp.n1; out[5] = sp.halo;
    constexpr int SUB = kSub, TILE = 64 * SUB, NW = 8;
    const long long C = (long long)NW * TILE - 2LL * sp.halo;
    if (C < 1024) { out[1] = kSlowMixing; return 0; }
    const long long nwg = (T - fp.nhs + C - 1) / C;
    out[6] = (double)nwg;
    double mu_end[D], quad = 0.0;
    smooth_head_forward<D>(m, sp, y, mu_end, &quad, hh_t);
    if (!smooth_head_tables<D>(m, sp)) { out[1] = kNotPD; return 0; }
    double rU[D * D], rv0[D], rs0 = 0.0;
    if (rnd && !smooth_rand_factors<D>(sp, eps_0, rU, rv0, &rs0)) { out[1] = kNotPD; return 0; }
    static thread_local Tables<D> tb;
    lane_table<D>(fp.P, tb.pw);
    lane_table<D>(sp.GP, tb.pg);
    double xi_out[D] = {0.0}, ssq = 0.0;
    std::vector<double> u(NW * 64 * SUB), r(NW * 64 * SUB), o0(NW * 64 * SUB);
    for (long long g = 0; g < nwg; ++g) {
        const long long c_lo = fp.nhs + g * C, c_hi = (c_lo + C < T) ? c_lo + C : T;
        const bool first = g == 0;
        const bool from_head = first || c_lo - sp.halo < fp.nhs;      // (a span shorter than a halo: the second workgroup starts behind the head as well)
        const long long s0 = from_head ? fp.nhs : c_lo - sp.halo;
        double sF[NW][D], sB[NW][D], X[NW][64][D], ST[NW][64][D], XI[NW][64][D];
        bool valid[NW];
        for (int w = 0; w < NW; ++w) {
            const long long tile_t0 = s0 + (long long)w * TILE;
            valid[w] = tile_t0 < T && tile_t0 < c_hi + sp.halo;
            for (int l = 0; l < 64; ++l) {
                const long long t0 = tile_t0 + (long long)l * SUB;
                double x[D];
                for (int i = 0; i < D; ++i) x[i] = 0.0;
                for (int j = 0; j < SUB; ++j) {
                    const int e = (w * 64 + l) * SUB + j;
                    u[e] = (valid[w] && t0 + j < T) ? y[t0 + j] - (hh_t ? hh_t[t0 + j] : fp.hh) : 0.0;
                    r[e] = 0.0;
                    if (!valid[w]) continue;
                    double rr = u[e];
                    for (int k = 0; k < D; ++k) rr = std::fma(-fp.h[k], x[k], rr);
                    r[e] = rr;
                    double nx[D];
                    for (int i = 0; i < D; ++i) {
                        double v = std::fma(fp.kA[i], u[e], fp.a[i]);
                        for (int k = 0; k < D; ++k) v = std::fma(fp.Phi[i * D + k], x[k], v);
                        nx[i] = v;
                    }
                    for (int i = 0; i < D; ++i) x[i] = nx[i];
                }
                for (int i = 0; i < D; ++i) X[w][l][i] = x[i];
            }
            if (valid[w]) {
                for (int K = 0; K < 4; ++K) {      // row_shr levels
                    double nx[64][D];
                    for (int l = 0; l < 64; ++l) {
                        for (int i = 0; i < D; ++i) nx[l][i] = X[w][l][i];
                        if ((l & 15) >= (1 << K)) matvec_acc<D>(fp.P[K], X[w][l - (1 << K)], nx[l]);
                    }
                    std::memcpy(X[w], nx, sizeof nx);
                }
                {      // row_bcast:15 into rows 1 and 3
                    double nx[64][D];
                    for (int l = 0; l < 64; ++l) {
                        for (int i = 0; i < D; ++i) nx[l][i] = X[w][l][i];
                        const int row = l >> 4;
                        if (row == 1 || row == 3) matvec_acc2<D>(tb.pw[(l & 15) + 1], X[w][16 * row - 1], nx[l]);
                    }
                    std::memcpy(X[w], nx, sizeof nx);
                }
                {      // row_bcast:31 into rows 2 and 3
                    double nx[64][D];
                    for (int l = 0; l < 64; ++l) {
                        for (int i = 0; i < D; ++i) nx[l][i] = X[w][l][i];
                        if (l >= 32) matvec_acc2<D>(tb.pw[l - 31], X[w][31], nx[l]);
                    }
                    std::memcpy(X[w], nx, sizeof nx);
                }
            }
            for (int l = 0; l < 64; ++l)
                for (int i = 0; i < D; ++i) ST[w][l][i] = l > 0 ? X[w][l - 1][i] : 0.0;
            for (int i = 0; i < D; ++i) sF[w][i] = valid[w] ? X[w][63][i] : 0.0;
        }
        for (int w = 0; w < NW; ++w) {
            const long long tile_t0 = s0 + (long long)w * TILE;
            for (int l = 0; l < 64; ++l)
                for (int i = 0; i < D; ++i) XI[w][l][i] = 0.0;
            if (!valid[w]) continue;
            double zin[D];
            for (int i = 0; i < D; ++i) zin[i] = 0.0;
            for (int k = 1; k <= 3; ++k) {
                const int src = w - k;
                if (src < -1 || (src == -1 && !from_head)) continue;
                const double* xs = src >= 0 ? sF[src] : mu_end;
                if (k == 1) for (int i = 0; i < D; ++i) zin[i] += xs[i];
                else matvec_acc<D>(fp.PT[k - 2], xs, zin);
            }
            for (int l = 0; l < 64; ++l) {
                const long long t0 = tile_t0 + (long long)l * SUB;
                double st[D];
                for (int i = 0; i < D; ++i) st[i] = ST[w][l][i];
                matvec_acc2<D>(tb.pw[l], zin, st);
                for (int j = 0; j < SUB; ++j) {
                    const int e = (w * 64 + l) * SUB + j;
                    double rr = r[e];
                    for (int k = 0; k < D; ++k) rr = std::fma(-sp.WJ[j][k], st[k], rr);
                    const long long t = t0 + j;
                    rr = t < T ? rr : 0.0;
                    r[e] = rr;
                    if (t >= c_lo && t < c_hi) ssq = std::fma(rr, rr, ssq);
                }
                double xi[D];
                for (int i = 0; i < D; ++i) xi[i] = 0.0;
                for (int j = SUB - 1; j >= 0; --j) {
                    const int e = (w * 64 + l) * SUB + j;
                    double o = 0.0;
                    for (int k = 0; k < D; ++k) o = std::fma(fp.h[k], xi[k], o);
                    o0[e] = o;
                    double np[D];
                    for (int i = 0; i < D; ++i) {
                        double v = sp.c[i] * r[e];
                        for (int k = 0; k < D; ++k) v = std::fma(sp.G[i * D + k], xi[k], v);
                        if (rnd) {
                            const long long t = t0 + j;
                            for (int k = 0; k <= i; ++k) v = std::fma(rU[k * D + i], t < T ? eps_t[t * D + k] : 0.0, v);
                            if (t == T - 1) v += rv0[i];
                        }
                        np[i] = v;
                    }
                    for (int i = 0; i < D; ++i) xi[i] = np[i];
                }
                for (int i = 0; i < D; ++i) XI[w][l][i] = xi[i];
            }
            for (int K = 0; K < 4; ++K) {      // row_shl levels
                double nx[64][D];
                for (int l = 0; l < 64; ++l) {
                    for (int i = 0; i < D; ++i) nx[l][i] = XI[w][l][i];
                    if ((l & 15) + (1 << K) <= 15) matvec_acc<D>(sp.GP[K], XI[w][l + (1 << K)], nx[l]);
                }
                std::memcpy(XI[w], nx, sizeof nx);
            }
            {      // rows 0 and 2 take the first lane of the row above them
                double nx[64][D];
                for (int l = 0; l < 64; ++l) {
                    for (int i = 0; i < D; ++i) nx[l][i] = XI[w][l][i];
                    const int row = l >> 4;
                    if (row == 0 || row == 2) matvec_acc2<D>(tb.pg[16 - (l & 15)], XI[w][16 * (row + 1)], nx[l]);
                }
                std::memcpy(XI[w], nx, sizeof nx);
            }
            {      // the lower half takes lane 32
                double nx[64][D];
                for (int l = 0; l < 64; ++l) {
                    for (int i = 0; i < D; ++i) nx[l][i] = XI[w][l][i];
                    if (l < 32) matvec_acc2<D>(tb.pg[32 - l], XI[w][32], nx[l]);
                }
                std::memcpy(XI[w], nx, sizeof nx);
            }
        }
        for (int w = 0; w < NW; ++w)
            for (int i = 0; i < D; ++i) sB[w][i] = valid[w] ? XI[w][0][i] : 0.0;
        for (int w = 0; w < NW; ++w) {
            if (!valid[w]) continue;
            const long long tile_t0 = s0 + (long long)w * TILE;
            double zin[D];
            for (int i = 0; i < D; ++i) zin[i] = 0.0;
            for (int k = 1; k <= 3; ++k) {
                const int src = w + k;
                if (src >= NW) continue;
                if (k == 1) for (int i = 0; i < D; ++i) zin[i] += sB[src][i];
                else matvec_acc<D>(sp.GPT[k - 2], sB[src], zin);
            }
            if (first && w == 0) {
                for (int i = 0; i < D; ++i) xi_out[i] = XI[0][0][i];
                matvec_acc<D>(sp.GPT[0], zin, xi_out);
            }
            for (int l = 0; l < 64; ++l) {
                const long long t0 = tile_t0 + (long long)l * SUB;
                double pin[D];
                for (int i = 0; i < D; ++i) pin[i] = l < 63 ? XI[w][l + 1][i] : 0.0;
                matvec_acc2<D>(tb.pg[63 - l], zin, pin);
                for (int j = 0; j < SUB; ++j) {
                    const long long t = t0 + j;
                    if (t < c_lo || t >= c_hi) continue;
                    const int e = (w * 64 + l) * SUB + j;
                    double o = u[e] + (hh_t ? hh_t[t] : fp.hh) - sp.rS * r[e] + o0[e];
                    for (int k = 0; k < D; ++k) o = std::fma(sp.WG[j][k], pin[k], o);
                    if (rnd) {
                        if (t == T - 1) o += rs0;
                        mean[t] = std::fma(std::sqrt(rnew_per_step ? Rnew[t] : Rnew[0]), eps_e[t], o);
                        continue;
                    }
                    mean[t] = o;
                    double v = sp.vb;
                    if (T - 1 - t < sp.n1) v = tvb[T - 1 - t];
                    var[t] = v + (rnew_per_step ? Rnew[t] : Rnew[0]);
                }
            }
        }
    }
    std::vector<double> hm(fp.nhs), hv(fp.nhs);
    if (rnd) {
        smooth_head_backward_rand<D>(m, sp, y, xi_out, eps_e, eps_t, Rnew, rnew_per_step != 0, hm.data());
        for (int t = 0; t < fp.nhs; ++t) mean[t] = hm[t];
        out[0] = 0.0;
        return 0;
    }
    smooth_head_backward<D>(m, sp, y, xi_out, hm.data(), hv.data());
    for (int t = 0; t < fp.nhs; ++t) {
        mean[t] = hm[t];
        var[t] = hv[t] + (rnew_per_step ? Rnew[t] : Rnew[0]);
    }
    out[0] = -0.5 * ((double)T * 1.8378770664093454835606594728112 + fp.LS + (double)(T - fp.n0) * fp.logS + quad + fp.iS * ssq);
    return 0;
}

}  // namespace

extern "C" int smoothsim_run(int d, const double* A, const double* a, const double* Q, const double* H, double hh, double R, const double* x0m,
                             const double* x0P, long long T, const double* y, const double* Rnew, int rnew_per_step, double* mean, double* var,
                             double* out /*[8]: lml, why, n0, nhs, n1, halo, workgroups*/, const double* eps_t, const double* eps_e, const double* eps_0, const double* hh_t) {
    ModelHost m;
    m.d = d;
    m.A = A; m.a = a; m.Q = Q; m.H = H; m.hh = &hh; m.R = &R; m.x0m = x0m; m.x0P = x0P;
    for (int i = 0; i < 8; ++i) out[i] = 0.0;
    switch (d) {
        case 1: return run_d<1>(m, T, y, Rnew, rnew_per_step, mean, var, out, eps_t, eps_e, eps_0, hh_t);
        case 2: return run_d<2>(m, T, y, Rnew, rnew_per_step, mean, var, out, eps_t, eps_e, eps_0, hh_t);
        case 3: return run_d<3>(m, T, y, Rnew, rnew_per_step, mean, var, out, eps_t, eps_e, eps_0, hh_t);
        case 4: return run_d<4>(m, T, y, Rnew, rnew_per_step, mean, var, out, eps_t, eps_e, eps_0, hh_t);
        case 5: return run_d<5>(m, T, y, Rnew, rnew_per_step, mean, var, out, eps_t, eps_e, eps_0, hh_t);
        case 6: return run_d<6>(m, T, y, Rnew, rnew_per_step, mean, var, out, eps_t, eps_e, eps_0, hh_t);
        case 7: return run_d<7>(m, T, y, Rnew, rnew_per_step, mean, var, out, eps_t, eps_e, eps_0, hh_t);
        case 8: return run_d<8>(m, T, y, Rnew, rnew_per_step, mean, var, out, eps_t, eps_e, eps_0, hh_t);
    }
    return 1;
}
