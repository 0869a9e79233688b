// Host half of the sweep engine (tgp_sweep.hpp): the geometry of a call -- chunk length, warm-up lengths -- and the model's shared blocks in the
// form the kernel reads them.  Plain C++ (no HIP): tests/hostsim/sweepsim.cpp builds the same plan on the CPU.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>

#include "tgp_sweep_body.hpp"

namespace tgp_sweep {

constexpr int kMaxD = 4;
constexpr int kOwned = 62;           // chunks a wave owns (lanes 1 .. 62)

// Host copies of the model's shared blocks (column-major, as tgp_model_set takes them).
struct ModelHost {
    int d = 0;
    bool sde = false;
    const double* A = nullptr;      // LTI: [d*d];  SDE: the first transition A1
    const double* a = nullptr;      // [d]
    const double* Q = nullptr;      // LTI: [d*d];  SDE: Q1 (checked against Pinf - A1 Pinf A1')
    const double* H = nullptr;      // [d]
    double hh = 0.0, R = 0.0;       // shared values (ignored where a per-step stream is given; R also feeds the warm-up estimate)
    const double* x0m = nullptr;    // [d]
    const double* x0P = nullptr;    // [d*d]  (SDE: also Pinf)
    const double* coef = nullptr;   // SDE: [lambda d | N d*d | N^2/2 d*d | ...] as ModelView::sde
    double tau_typ = 0.0;           // SDE: a typical gap (the warm-up estimate)
};

struct Plan {
    int d = 0;
    bool sde = false;
    int C = 0, W = 0, Wb = 0;
    int64_t T = 0, nchunks = 0, nwaves = 0;
    double mc[160];                 // the ModelC<d> of the call, as plain doubles (copied into the typed kernel argument)
};
struct Forced {                     // test hook: geometry of the next plan (0: automatic)
    int C = 0, W = 0, Wb = 0;
};

namespace plan_detail {

template <int D> inline void fill_model(const ModelHost& m, ModelC<D>& mc) {
    std::memset(&mc, 0, sizeof mc);
    for (int i = 0; i < D * D; ++i) mc.A[i] = m.A[i];
    for (int i = 0; i < D; ++i) {
        mc.a[i] = m.a[i];
        mc.H[i] = m.H[i];
        mc.x0m[i] = m.x0m[i];
    }
    for (int j = 0; j < D; ++j)
        for (int i = 0; i <= j; ++i) {
            mc.Q[pidx(i, j)] = 0.5 * (m.Q[i + j * D] + m.Q[j + i * D]);
            mc.x0P[pidx(i, j)] = m.x0P[i + j * D];      // Symmetric(P): the upper triangle is what the reference reads (lgc.jl:50)
        }
    mc.hh = m.hh;
    mc.R = m.R;
    if (m.sde) {
        const double* q = m.coef;
        for (int i = 0; i < D; ++i) mc.lam[i] = q[i];
        for (int i = 0; i < D * D; ++i) {
            mc.N1[i] = q[D + i];
            mc.N2[i] = q[D + D * D + i];
        }
    }
}

// stationary covariance of x' = A x + w, w ~ N(0, Q): P = A P A' + Q by doubling (spectral radius < 1); false: no such P
template <int D> inline bool lyapunov(const double* A, const double* Q, double* P) {
    double M[D * D], S[D * D], T1[D * D], T2[D * D];
    for (int i = 0; i < D * D; ++i) { M[i] = A[i]; S[i] = Q[i]; }
    for (int it = 0; it < 60; ++it) {
        // S <- S + M S M' ; M <- M M
        for (int j = 0; j < D; ++j)
            for (int i = 0; i < D; ++i) {
                double a = 0.0;
                for (int k = 0; k < D; ++k) a += M[i + k * D] * S[k + j * D];
                T1[i + j * D] = a;
            }
        double dmax = 0.0, smax = 0.0;
        for (int j = 0; j < D; ++j)
            for (int i = 0; i < D; ++i) {
                double a = 0.0;
                for (int k = 0; k < D; ++k) a += T1[i + k * D] * M[j + k * D];
                T2[i + j * D] = a;
                dmax = std::max(dmax, std::fabs(a));
            }
        for (int i = 0; i < D * D; ++i) { S[i] += T2[i]; smax = std::max(smax, std::fabs(S[i])); }
        for (int j = 0; j < D; ++j)
            for (int i = 0; i < D; ++i) {
                double a = 0.0;
                for (int k = 0; k < D; ++k) a += M[i + k * D] * M[k + j * D];
                T1[i + j * D] = a;
            }
        for (int i = 0; i < D * D; ++i) M[i] = T1[i];
        if (!std::isfinite(smax) || smax > 1e200) return false;
        if (dmax <= 1e-17 * smax) {
            for (int i = 0; i < D * D; ++i) P[i] = S[i];
            return true;
        }
    }
    return false;
}

// How many steps until a start state is forgotten to `tol`: the closed loop of the stationary filter with every step observed at noise R,
// Phi = A (I - K H'); the number of steps n with max |Phi^n| <= tol (a Jordan block's powers carry a polynomial factor the spectral
// radius does not show: multiply, do not estimate).  0: not within `cap` steps.
template <int D> inline int forget_steps(const double* A, const double* Q, const double* H, double R, double tol, int cap) {
    double P[D * D];
    if (!lyapunov<D>(A, Q, P)) return 0;
    double Phi[D * D];
    for (int it = 0; it < 4096; ++it) {      // Riccati iteration to the stationary PREDICTED covariance
        double V[D], s = R;
        for (int j = 0; j < D; ++j) {
            double a = 0.0;
            for (int k = 0; k < D; ++k) a += H[k] * P[k + j * D];
            V[j] = a;
        }
        for (int k = 0; k < D; ++k) s += V[k] * H[k];
        if (!(s > 0.0)) return 0;
        double Pf[D * D], AP[D * D], Pn[D * D];
        for (int j = 0; j < D; ++j)
            for (int i = 0; i < D; ++i) Pf[i + j * D] = P[i + j * D] - V[i] * V[j] / s;
        for (int j = 0; j < D; ++j)
            for (int i = 0; i < D; ++i) {
                double a = 0.0;
                for (int k = 0; k < D; ++k) a += A[i + k * D] * Pf[k + j * D];
                AP[i + j * D] = a;
            }
        double ch = 0.0, sc = 0.0;
        for (int j = 0; j < D; ++j)
            for (int i = 0; i < D; ++i) {
                double a = Q[i + j * D];
                for (int k = 0; k < D; ++k) a += AP[i + k * D] * A[j + k * D];
                Pn[i + j * D] = a;
                ch = std::max(ch, std::fabs(a - P[i + j * D]));
                sc = std::max(sc, std::fabs(a));
            }
        for (int i = 0; i < D * D; ++i) P[i] = Pn[i];
        if (ch <= 1e-14 * sc || it == 4095) {
            // Phi = A (I - K H'), K = V / s
            for (int j = 0; j < D; ++j)
                for (int i = 0; i < D; ++i) {
                    double a = A[i + j * D];
                    double ak = 0.0;
                    for (int k = 0; k < D; ++k) ak += A[i + k * D] * V[k];
                    a -= ak / s * H[j];
                    Phi[i + j * D] = a;
                }
            break;
        }
    }
    // scale-free powers: in the coordinates of the stationary prior's standard deviations
    double sd[D];
    {
        double Ps[D * D];
        if (!lyapunov<D>(A, Q, Ps)) return 0;
        for (int i = 0; i < D; ++i) sd[i] = std::sqrt(std::max(Ps[i + i * D], 1e-300));
    }
    double M[D * D], Tm[D * D];
    for (int j = 0; j < D; ++j)
        for (int i = 0; i < D; ++i) M[i + j * D] = Phi[i + j * D] * sd[j] / sd[i];
    double Pw[D * D];
    std::memcpy(Pw, M, sizeof Pw);
    for (int n = 1; n <= cap; ++n) {
        double mx = 0.0;
        for (int i = 0; i < D * D; ++i) mx = std::max(mx, std::fabs(Pw[i]));
        if (mx <= tol) return n;
        for (int j = 0; j < D; ++j)
            for (int i = 0; i < D; ++i) {
                double a = 0.0;
                for (int k = 0; k < D; ++k) a += Pw[i + k * D] * M[k + j * D];
                Tm[i + j * D] = a;
            }
        std::memcpy(Pw, Tm, sizeof Pw);
    }
    return 0;
}

template <int D> inline void host_sde_A(const double* q, double tau, double* A) {
    for (int j = 0; j < D; ++j)
        for (int i = 0; i < D; ++i)
            A[i + j * D] = std::exp(-q[i] * tau) * (tau * tau * q[D + D * D + i + j * D] + tau * q[D + i + j * D] + (i == j ? 1.0 : 0.0));
}

// What a handed-over state may be off by, relative to the size of a state.  Forwards 1e-12: the log marginal likelihood is held to 1e-10 relative
// and the innovations behind a hand-over inherit its error.  Backwards 1e-11: the marginals are held to 1e-8 of their scale.
constexpr double kTol = 1e-12, kTolBack = 1e-11;
constexpr int kMaxWarm = 4096;

template <int D> inline bool plan_d(Plan* e, const Forced& f, const ModelHost& m, int64_t T, int w_hint, int wb_hint, int num_cu, std::string* why) {
    constexpr int B = Geo<D>::B;
    ModelC<D> mc;
    fill_model<D>(m, mc);
    mc.tol = kTol;
    mc.tol_b = kTolBack;
    double Aty[D * D], Qty[D * D];      // a typical step's transition (the warm-up estimate)
    if (m.sde) {
        // the warm-up starts from Pinf = x0P; the first transition must be consistent with it (Q1 = Pinf - A1 Pinf A1')
        double Pinf[D * D];
        for (int j = 0; j < D; ++j)
            for (int i = 0; i < D; ++i) Pinf[i + j * D] = 0.5 * (m.x0P[i + j * D] + m.x0P[j + i * D]);
        double worst = 0.0, scale = 0.0;
        for (int j = 0; j < D; ++j)
            for (int i = 0; i < D; ++i) {
                double a = 0.0;
                for (int k = 0; k < D; ++k)
                    for (int l = 0; l < D; ++l) a += m.A[i + k * D] * Pinf[k + l * D] * m.A[j + l * D];
                worst = std::max(worst, std::fabs(Pinf[i + j * D] - a - m.Q[i + j * D]));
                scale = std::max(scale, std::fabs(Pinf[i + j * D]));
            }
        if (!(worst <= 1e-11 * scale)) {
            if (why) *why = "first transition not of the form Q1 = Pinf - A1 Pinf A1'";
            return false;
        }
        for (int i = 0; i < D; ++i) mc.gm[i] = m.x0m[i];
        for (int j = 0; j < D; ++j)
            for (int i = 0; i <= j; ++i) mc.gP[pidx(i, j)] = Pinf[i + j * D];
        host_sde_A<D>(m.coef, m.tau_typ, Aty);
        for (int j = 0; j < D; ++j)
            for (int i = 0; i < D; ++i) {
                double a = 0.0;
                for (int k = 0; k < D; ++k)
                    for (int l = 0; l < D; ++l) a += Aty[i + k * D] * Pinf[k + l * D] * Aty[j + l * D];
                Qty[i + j * D] = Pinf[i + j * D] - a;
            }
    } else {
        for (int i = 0; i < D * D; ++i) { Aty[i] = m.A[i]; Qty[i] = 0.5 * (m.Q[i] + m.Q[(i % D) * D + i / D]); }
        double Ps[D * D];
        if (!lyapunov<D>(Aty, Qty, Ps)) {
            if (why) *why = "the transition has no stationary covariance";
            return false;
        }
        // stationary mean: (I - A) gm = a, by fixed-point iteration (spectral radius < 1)
        double gm[D] = {};
        for (int it = 0; it < 100000; ++it) {
            double nx[D], ch = 0.0;
            for (int i = 0; i < D; ++i) {
                double acc = m.a[i];
                for (int k = 0; k < D; ++k) acc += Aty[i + k * D] * gm[k];
                nx[i] = acc;
                ch = std::max(ch, std::fabs(acc - gm[i]));
            }
            std::memcpy(gm, nx, sizeof gm);
            if (ch == 0.0 || ch < 1e-300) break;
            double mx = 0.0;
            for (int i = 0; i < D; ++i) mx = std::max(mx, std::fabs(gm[i]));
            if (ch <= 1e-15 * mx) break;
        }
        for (int i = 0; i < D; ++i) mc.gm[i] = gm[i];
        for (int j = 0; j < D; ++j)
            for (int i = 0; i <= j; ++i) mc.gP[pidx(i, j)] = 0.5 * (Ps[i + j * D] + Ps[j + i * D]);
    }
    int W = w_hint, Wb = wb_hint;
    if (W <= 0 || Wb <= 0) {
        // (the estimate is for a series observed at every step with the representative noise: missing steps and larger noise slow the forgetting
        //  -- hence the margin -- and the checks decide: a call that finds its warm-ups short is repeated with longer ones and the bound model
        //  remembers them.  Measured at the bench model with 10 % missing: 1e-12 forwards needs 100 steps, the estimate says 96 + margin.)
        const double Rrep = m.R > 0.0 && m.R < 1e14 ? m.R : 1.0;
        const int nf = forget_steps<D>(Aty, Qty, m.H, Rrep, kTol, kMaxWarm);
        const int nb = forget_steps<D>(Aty, Qty, m.H, Rrep, kTolBack, kMaxWarm);
        if (nf == 0 || nb == 0) {
            if (why) *why = "the filter does not forget a state within 4096 steps";
            return false;
        }
        if (W <= 0) W = nf + nf / 8 + 8;
        if (Wb <= 0) Wb = nb + nb / 8 + 8;
    }
    auto up = [](int v, int q) { return (v + q - 1) / q * q; };
    W = up(W, 8);
    Wb = up(Wb, 8);
    // chunk length: the machine holds 4 waves per CU (their LDS), every wave 62 chunks of its own; a chunk holds the backward warm-up
    const int64_t slots = (int64_t)std::max(num_cu, 1) * 4 * kOwned;
    int64_t C = up((int)std::min<int64_t>((T + slots - 1) / slots, 1 << 20), 8);
    C = std::max<int64_t>(C, 64);
    C = std::max<int64_t>(C, Wb);
    if (f.C > 0) C = up(f.C, 8);
    if (f.W > 0) W = up(f.W, 8);
    if (f.Wb > 0) Wb = up(f.Wb, 8);
    if (Wb > C) Wb = (int)C;
    if (W > kMaxWarm || (f.C == 0 && W > 4 * C)) {
        if (why) *why = "forward warm-up longer than four chunks";
        return false;
    }
    (void)B;
    e->d = D;
    e->sde = m.sde;
    e->T = T;
    e->C = (int)C;
    e->W = W;
    e->Wb = Wb;
    e->nchunks = (T + C - 1) / C;
    e->nwaves = (e->nchunks + kOwned - 1) / kOwned;
    static_assert(sizeof(ModelC<D>) <= sizeof(e->mc), "Plan::mc too small");
    std::memcpy(e->mc, &mc, sizeof mc);
    return true;
}

}  // namespace plan_detail

// Chooses the geometry (chunk length, warm-ups) for this model and series; false: the engine declines (`why` says so).
// w_hint / wb_hint: warm-ups a previous call on the same bound model needed (0: estimate from the model).
inline bool make_plan(Plan* p, const Forced& f, const ModelHost& m, int64_t T, int w_hint, int wb_hint, int num_cu, std::string* why) {
    if (m.d < 1 || m.d > kMaxD || T < 1) {
        if (why) *why = "state dimension";
        return false;
    }
    switch (m.d) {
        case 1: return plan_detail::plan_d<1>(p, f, m, T, w_hint, wb_hint, num_cu, why);
        case 2: return plan_detail::plan_d<2>(p, f, m, T, w_hint, wb_hint, num_cu, why);
        case 3: return plan_detail::plan_d<3>(p, f, m, T, w_hint, wb_hint, num_cu, why);
        default: return plan_detail::plan_d<4>(p, f, m, T, w_hint, wb_hint, num_cu, why);
    }
}

}  // namespace tgp_sweep
