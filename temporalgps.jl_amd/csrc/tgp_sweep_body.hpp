// Per-lane arithmetic of the sweep engine (tgp_sweep.hip): the reference's sequential recursions for ONE chunk of consecutive
// steps with time-varying gains -- a missing-data mask, a noise variance per step, irregular spacing -- on a state (m, packed upper P) in
// registers.  Host- and device-callable: tests/hostsim runs the very same functions lane by lane on the CPU (tests/test_sweep_host.py).
//
// Reference semantics restated here (file:line under /root/reference/src):
//   predict              models/linear_gaussian_conditionals.jl:46-52     (A Symmetric(P)) A' + Q  (upper triangle kept)
//   posterior_and_lml    models/linear_gaussian_conditionals.jl:247-257   ScalarOutputLGC; a missing step is skipped (missings.jl:8-23:
//                        y := 0, R := 1e15 and the compensated volume term agree with skipping to ~1e-15 relative)
//   invert_dynamics      models/lgssm.jl:231-238                           G = Pf A' (Pp + 1e-10 I)^-1 by Cholesky, as there
//   step_marginals       models/lgssm.jl:111-115 (Reverse)                 x <- G x + g, P <- G P G' + L, in gain form:
//                        ms = mf + G (ms+ - mp),  Ps = Pf + G (Ps+ - Pp - 1e-10 I) G'   (L = Pf - G (Pp + 1e-10 I) G' substituted)
//   broadcast_components gp/lti_sde.jl:135-146                             A_k = exp(F tau_k) in closed form per Matern block;
//                        Q_k = Pinf - A_k Pinf A_k' is never formed: A P A' + Q_k = A (P - Pinf) A' + Pinf
#pragma once
#include "tgp_math.hpp"

namespace tgp_sweep {

constexpr double kJitter = 1e-10;      // lgssm.jl:235
constexpr double kLog2Pi = 1.8378770664093454835606594728112;

template <int D> struct SD {
    static constexpr int DD = D * D, DS = D * (D + 1) / 2, NS = D + DS;
};
// packed upper triangle, column by column (tgp_math_body.inc store_sym): entry (i, j), i <= j, sits at j (j + 1) / 2 + i
TGP_HD constexpr int pidx(int i, int j) { return i <= j ? j * (j + 1) / 2 + i : i * (i + 1) / 2 + j; }

// 1 / x and (sqrt x, 1 / sqrt x) from the hardware's approximate v_rcp_f64 / v_rsq_f64 and two Newton (Goldschmidt) steps each: ~1 ulp,
// 5 / 8 instructions instead of the 12 / 25 of an IEEE division / sqrt + division (the passes are bound by their instruction stream).
TGP_HD double fast_rcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(x);
    double e = ::fma(-x, r, 1.0);
    r = ::fma(r, e, r);
    e = ::fma(-x, r, 1.0);
    r = ::fma(r, e, r);
    return r;
#else
    return 1.0 / x;
#endif
}
TGP_HD void fast_sqrt_rsqrt(double x, double& s, double& rs) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double r0 = __builtin_amdgcn_rsq(x);
    double g = x * r0, hh = 0.5 * r0;
    double e = ::fma(-hh, g, 0.5);
    g = ::fma(g, e, g);
    hh = ::fma(hh, e, hh);
    e = ::fma(-hh, g, 0.5);
    g = ::fma(g, e, g);
    hh = ::fma(hh, e, hh);
    s = g;
    rs = hh + hh;
#else
    s = ::sqrt(x);
    rs = 1.0 / s;
#endif
}

// ---- the model's shared blocks, as the kernel reads them from its argument segment (scalar loads) ----------------------------------
template <int D> struct ModelC {
    double A[D * D], a[D], Q[SD<D>::DS];      // LTI transition.  SDE: A = A1, the series' FIRST transition (lti_sde.jl:139; its Q1 must be
                                              // Pinf - A1 Pinf A1', which the host checks), Q unused
    double H[D];
    double x0m[D], x0P[SD<D>::DS];            // the prior of the series
    double gm[D], gP[SD<D>::DS];              // where a warm-up starts from: the stationary prior (SDE: Pinf)
    double lam[D], N1[D * D], N2[D * D];      // SDE: exp(F tau) = e^(-lam tau) (I + tau N1 + tau^2 N2) per block (ModelView::sde)
    double hh, R;                             // shared emission offset / noise variance (used where no per-step stream is given)
    double tol, tol_b;                        // relative size of a forgotten start state: the checks of the forward / backward hand-overs
};

template <int D> struct State {
    double m[D];
    double P[SD<D>::DS];
};

template <int D> TGP_HD void set_state(State<D>& x, const double* m, const double* P) {
    TGP_UNROLL for (int i = 0; i < D; ++i) x.m[i] = m[i];
    TGP_UNROLL for (int i = 0; i < SD<D>::DS; ++i) x.P[i] = P[i];
}

// A = exp(F tau) in closed form (tgp_chunk.hpp sde_transition_impl, without its Q)
template <int D> TGP_HD void sde_A(const ModelC<D>& mc, double tau, double* A) {
    double e[D];
    e[0] = ::exp(-mc.lam[0] * tau);
    TGP_UNROLL for (int i = 1; i < D; ++i) e[i] = (mc.lam[i] == mc.lam[i - 1]) ? e[i - 1] : ::exp(-mc.lam[i] * tau);     // (wave-uniform: one exp per block)
    const double t2 = tau * tau;
    TGP_UNROLL for (int j = 0; j < D; ++j)
        TGP_UNROLL for (int i = 0; i < D; ++i) A[i + j * D] = e[i] * ::fma(t2, mc.N2[i + j * D], ::fma(tau, mc.N1[i + j * D], i == j ? 1.0 : 0.0));
}

// AX = A Symmetric(X) (full d x d), X packed
template <int D> TGP_HD void mul_A_sym(const double* A, const double* X, double* AX) {
    TGP_UNROLL for (int j = 0; j < D; ++j)
        TGP_UNROLL for (int i = 0; i < D; ++i) {
            double acc = 0.0;
            TGP_UNROLL for (int k = 0; k < D; ++k) acc = ::fma(A[i + k * D], X[pidx(k, j)], acc);
            AX[i + j * D] = acc;
        }
}
// out (packed upper) = AX A' + C (packed)
template <int D> TGP_HD void mul_AXAt_plus(const double* AX, const double* A, const double* C, double* out) {
    TGP_UNROLL for (int j = 0; j < D; ++j)
        TGP_UNROLL for (int i = 0; i <= j; ++i) {
            double acc = C[pidx(i, j)];
            TGP_UNROLL for (int k = 0; k < D; ++k) acc = ::fma(AX[i + k * D], A[j + k * D], acc);
            out[pidx(i, j)] = acc;
        }
}

// The transition of one step, in the form the predict needs: LTI -> (A, a, Q) of the model; SDE -> A from tau, covariance through Pinf.
template <int D, bool SDE> struct Trans {
    double A[D * D];
    TGP_HD void set(const ModelC<D>& mc, double tau, bool first) {
        if constexpr (SDE) {
            sde_A<D>(mc, first ? 0.0 : tau, A);
            TGP_UNROLL for (int i = 0; i < D * D; ++i) A[i] = first ? mc.A[i] : A[i];
        } else {
            TGP_UNROLL for (int i = 0; i < D * D; ++i) A[i] = mc.A[i];
        }
    }
};

// predict (lgc.jl:46-52).
// AP_out (optional, full d x d): A Symmetric(P) of the INCOMING covariance -- what invert_dynamics multiplies again (lgssm.jl:234).
template <int D, bool SDE> TGP_HD void predict(const ModelC<D>& mc, const double* A, double* m, double* P, double* AP_out = nullptr) {
    double mp[D];
    TGP_UNROLL for (int i = 0; i < D; ++i) {
        double acc = mc.a[i];
        TGP_UNROLL for (int k = 0; k < D; ++k) acc = ::fma(A[i + k * D], m[k], acc);
        mp[i] = acc;
    }
    TGP_UNROLL for (int i = 0; i < D; ++i) m[i] = mp[i];
    double AX[D * D], Pn[SD<D>::DS];
    if constexpr (SDE) {
        if (AP_out) {      // A P and A Pinf separately: the smoother needs A P itself
            double AG[D * D];
            mul_A_sym<D>(A, P, AP_out);
            mul_A_sym<D>(A, mc.gP, AG);
            TGP_UNROLL for (int i = 0; i < D * D; ++i) AX[i] = AP_out[i] - AG[i];
        } else {
            double Dm[SD<D>::DS];
            TGP_UNROLL for (int i = 0; i < SD<D>::DS; ++i) Dm[i] = P[i] - mc.gP[i];
            mul_A_sym<D>(A, Dm, AX);
        }
        mul_AXAt_plus<D>(AX, A, mc.gP, Pn);
    } else {
        mul_A_sym<D>(A, P, AX);
        if (AP_out) { TGP_UNROLL for (int i = 0; i < D * D; ++i) AP_out[i] = AX[i]; }
        mul_AXAt_plus<D>(AX, A, mc.Q, Pn);
    }
    TGP_UNROLL for (int i = 0; i < SD<D>::DS; ++i) P[i] = Pn[i];
}

// The log marginal likelihood's running sums of one lane: sum of r^2 / S, the product of the S of up to eight steps (ONE log per eight
// steps: log prod = sum log to rounding), the observed steps.
struct LmlAcc {
    double q = 0.0, logs = 0.0, prod = 1.0, n = 0.0;
    TGP_HD void flush() {
        logs += ::log(prod);
        prod = 1.0;
    }
    TGP_HD double total() const { return -0.5 * (n * kLog2Pi + logs + q); }
};

// posterior_and_lml of a ScalarOutputLGC (lgc.jl:247-257); obs == false: the step is missing, nothing changes.
template <int D> TGP_HD void update(const double* H, double hh, double R, double y, bool obs, double* m, double* P, LmlAcc* acc, bool& ok) {
    double V[D];
    TGP_UNROLL for (int j = 0; j < D; ++j) {
        double a = 0.0;
        TGP_UNROLL for (int k = 0; k < D; ++k) a = ::fma(H[k], P[pidx(k, j)], a);
        V[j] = a;
    }
    double s2 = R, hm = hh;
    TGP_UNROLL for (int k = 0; k < D; ++k) {
        s2 = ::fma(V[k], H[k], s2);
        hm = ::fma(H[k], m[k], hm);
    }
    const double S = obs ? s2 : 1.0;
    ok = ok && (S > 0.0);
    const double iS = obs ? fast_rcp(S) : 0.0;
    const double v = obs ? (y - hm) : 0.0;
    const double viS = v * iS;
    TGP_UNROLL for (int i = 0; i < D; ++i) m[i] = ::fma(V[i], viS, m[i]);
    TGP_UNROLL for (int j = 0; j < D; ++j) {
        const double w = V[j] * iS;
        TGP_UNROLL for (int i = 0; i <= j; ++i) P[pidx(i, j)] = ::fma(-V[i], w, P[pidx(i, j)]);
    }
    if (acc) {
        acc->q = ::fma(v, viS, acc->q);
        acc->prod *= S;
        acc->n += obs ? 1.0 : 0.0;
    }
}

// One step of the reverse-time recursion (lgssm.jl:231-238 + :111-115): xf = filtering state of step t, A = transition of step t + 1,
// xs = smoothing state of step t + 1 on entry, of step t on return.
template <int D, bool SDE> TGP_HD void smooth_step(const ModelC<D>& mc, const double* A, const State<D>& xf, State<D>& xs, bool& ok) {
    constexpr int DS = SD<D>::DS;
    double mp[D], Pp[DS], AP[D * D];
    TGP_UNROLL for (int i = 0; i < D; ++i) mp[i] = xf.m[i];
    TGP_UNROLL for (int i = 0; i < DS; ++i) Pp[i] = xf.P[i];
    predict<D, SDE>(mc, A, mp, Pp, AP);
    // upper Cholesky factor of Pp + jitter I (rows of U, reciprocal pivots)
    double U[D * D], inv[D];
    TGP_UNROLL for (int j = 0; j < D; ++j) {
        TGP_UNROLL for (int i = 0; i < j; ++i) {
            double acc = Pp[pidx(i, j)];
            TGP_UNROLL for (int k = 0; k < i; ++k) acc = ::fma(-U[k + i * D], U[k + j * D], acc);
            U[i + j * D] = acc * inv[i];
        }
        double acc = Pp[pidx(j, j)] + kJitter;
        TGP_UNROLL for (int k = 0; k < j; ++k) acc = ::fma(-U[k + j * D], U[k + j * D], acc);
        ok = ok && (acc > 0.0);
        double s, rs;
        fast_sqrt_rsqrt(acc, s, rs);
        U[j + j * D] = s;
        inv[j] = rs;
    }
    // W = (Pp + jitter I)^-1 (A Pf) = G'   (column j of W: U' z = AP[:, j], U w = z)
    double W[D * D];
    TGP_UNROLL for (int j = 0; j < D; ++j) {
        TGP_UNROLL for (int i = 0; i < D; ++i) {
            double acc = AP[i + j * D];
            TGP_UNROLL for (int k = 0; k < i; ++k) acc = ::fma(-U[k + i * D], W[k + j * D], acc);
            W[i + j * D] = acc * inv[i];
        }
        TGP_UNROLL for (int i = D - 1; i >= 0; --i) {
            double acc = W[i + j * D];
            TGP_UNROLL for (int k = i + 1; k < D; ++k) acc = ::fma(-U[i + k * D], W[k + j * D], acc);
            W[i + j * D] = acc * inv[i];
        }
    }
    // gain form: ms = mf + G (ms+ - mp);  Ps = Pf + G (Ps+ - Pp - jitter I) G'     (G[i][k] = W[k + i D])
    double dm[D], Dm[DS];
    TGP_UNROLL for (int i = 0; i < D; ++i) dm[i] = xs.m[i] - mp[i];
    TGP_UNROLL for (int j = 0; j < D; ++j)
        TGP_UNROLL for (int i = 0; i <= j; ++i) Dm[pidx(i, j)] = xs.P[pidx(i, j)] - Pp[pidx(i, j)] - (i == j ? kJitter : 0.0);
    TGP_UNROLL for (int i = 0; i < D; ++i) {
        double acc = xf.m[i];
        TGP_UNROLL for (int k = 0; k < D; ++k) acc = ::fma(W[k + i * D], dm[k], acc);
        xs.m[i] = acc;
    }
    double GD[D * D];
    TGP_UNROLL for (int j = 0; j < D; ++j)
        TGP_UNROLL for (int i = 0; i < D; ++i) {
            double acc = 0.0;
            TGP_UNROLL for (int k = 0; k < D; ++k) acc = ::fma(W[k + i * D], Dm[pidx(k, j)], acc);
            GD[i + j * D] = acc;
        }
    TGP_UNROLL for (int j = 0; j < D; ++j)
        TGP_UNROLL for (int i = 0; i <= j; ++i) {
            double acc = xf.P[pidx(i, j)];
            TGP_UNROLL for (int k = 0; k < D; ++k) acc = ::fma(GD[i + k * D], W[k + j * D], acc);
            xs.P[pidx(i, j)] = acc;
        }
}

// emission predict of a smoothing state (lgssm.jl:111-113 through lgc.jl:46-52 with the replaced noise): mean = H' m + h, var = H' P H + R'
template <int D> TGP_HD void emit(const double* H, double hh, double Rn, const State<D>& xs, double& mean, double& var) {
    double mu = hh, v = Rn;
    TGP_UNROLL for (int j = 0; j < D; ++j) {
        double acc = 0.0;
        TGP_UNROLL for (int k = 0; k < D; ++k) acc = ::fma(H[k], xs.P[pidx(k, j)], acc);
        v = ::fma(acc, H[j], v);
        mu = ::fma(H[j], xs.m[j], mu);
    }
    mean = mu;
    var = v;
}

// How far apart two states are, relative to the size of a state: what the checks of a forgotten start compare with mc.tol.
template <int D> TGP_HD double state_distance(const ModelC<D>& mc, const State<D>& a, const State<D>& b) {
    double worst = 0.0;
    TGP_UNROLL for (int i = 0; i < D; ++i) {
        const double sc = ::sqrt(mc.gP[pidx(i, i)]) + ::fabs(a.m[i]) + ::fabs(b.m[i]);
        const double r = ::fabs(a.m[i] - b.m[i]) / sc;
        worst = r > worst ? r : worst;      // (a NaN on either side compares false everywhere: caught by the caller's finite test)
    }
    TGP_UNROLL for (int j = 0; j < D; ++j)
        TGP_UNROLL for (int i = 0; i <= j; ++i) {
            const double sc = ::sqrt(mc.gP[pidx(i, i)] * mc.gP[pidx(j, j)]);
            const double r = ::fabs(a.P[pidx(i, j)] - b.P[pidx(i, j)]) / sc;
            worst = r > worst ? r : worst;
        }
    return worst;
}

// ---- the inputs of one step, as every pass sees them ------------------------------------------------------------------------------
struct Streams {
    const double* y = nullptr;
    const uint8_t* mask = nullptr;      // 1 = missing (nullptr: none)
    const double* R = nullptr;          // per step (nullptr: mc.R)
    const double* hh = nullptr;         // per step (nullptr: mc.hh)
    const double* tau = nullptr;        // SDE: gap to the previous time stamp (entry 0 unused: the first transition is explicit)
    const double* Rnew = nullptr;       // per step, or one value (rnew_per_step == 0)
    int rnew_per_step = 0;
};

// ---- the runs of one lane (the kernel of tgp_sweep.hip and tests/hostsim/sweepsim.cpp call the same code) ---------------------------------
template <int D> struct KArgs {
    ModelC<D> mc;
    Streams st;
    long long T;
    int C, W, Wb;               // steps per chunk, forward / backward warm-up: multiples of 8
    long long nchunks;
    double* mean;
    double* var;
    double* ckpt;               // [wave][block][component][lane]
    double* part;               // [wave][4]: share of the log marginal likelihood, status bits, forward / backward distance (pinned host memory)
};

// steps per block: a checkpoint of the forward run every B steps; the block's B filtering states sit in LDS during the backward run
template <int D> struct Geo {
    static constexpr int B = D <= 3 ? 8 : 4;
};

// Register residency of the model's shared blocks inside the runs.  Measured (scripts/ubench/fp64_issue.hip, one wave per SIMD): dependent
// v_fma_f64 issue back to back at 4.6 cycles, but a scalar load whose value is used right away costs ~100 cycles -- and the first version of
// this engine re-read its constants from the argument segment in every step (the compiler prefers re-loading an invariant value to keeping
// it).  A value that went through one of these statements is opaque to the optimiser: it stays in a register of the named file.
#if defined(__HIP_DEVICE_COMPILE__)
#define TGP_PIN_V(x) asm volatile("" : "+v"(x))
#define TGP_PIN_S(x) asm volatile("" : "+s"(x))
#else
#define TGP_PIN_V(x) ((void)0)
#define TGP_PIN_S(x) ((void)0)
#endif

// The model as the runs hold it: multipliers wave-uniform (scalar registers), addends in vector registers (an instruction takes ONE scalar
// operand).  SDE models: `A` is not kept (a step builds its own from tau; the series' first transition is read from the arguments where needed).
template <int D, bool SDE> struct ModelR {
    double A[SDE ? 1 : D * D];
    double a[D];
    double Qg[SD<D>::DS];                 // LTI: Q;  SDE: Pinf
    double H[D];
    double lam[SDE ? D : 1], N1[SDE ? D * D : 1], N2[SDE ? D * D : 1];
    double hh, R;
    TGP_HD void init(const ModelC<D>& mc) {
        if constexpr (!SDE) {
            TGP_UNROLL for (int i = 0; i < D * D; ++i) { A[i] = mc.A[i]; TGP_PIN_S(A[i]); }
            lam[0] = N1[0] = N2[0] = 0.0;
        } else {
            A[0] = 0.0;
            TGP_UNROLL for (int i = 0; i < D; ++i) { lam[i] = mc.lam[i]; TGP_PIN_S(lam[i]); }
            TGP_UNROLL for (int i = 0; i < D * D; ++i) { N1[i] = mc.N1[i]; N2[i] = mc.N2[i]; TGP_PIN_S(N1[i]); TGP_PIN_S(N2[i]); }
        }
        TGP_UNROLL for (int i = 0; i < D; ++i) { a[i] = mc.a[i]; H[i] = mc.H[i]; TGP_PIN_V(a[i]); TGP_PIN_S(H[i]); }
        TGP_UNROLL for (int i = 0; i < SD<D>::DS; ++i) { Qg[i] = SDE ? mc.gP[i] : mc.Q[i]; TGP_PIN_V(Qg[i]); }
        hh = mc.hh;
        R = mc.R;
        TGP_PIN_V(hh);
        TGP_PIN_V(R);
    }
};

// the transition matrix of step t (tau: the gap in front of it)
template <int D, bool SDE> TGP_HD void step_A(const ModelC<D>& mc, const ModelR<D, SDE>& mr, double tau, bool first, double* A) {
    if constexpr (SDE) {
        double e[D];
        e[0] = ::exp(-mr.lam[0] * tau);
        TGP_UNROLL for (int i = 1; i < D; ++i) e[i] = (mr.lam[i] == mr.lam[i - 1]) ? e[i - 1] : ::exp(-mr.lam[i] * tau);     // (wave-uniform: one exp per block)
        const double t2 = tau * tau;
        TGP_UNROLL for (int j = 0; j < D; ++j)
            TGP_UNROLL for (int i = 0; i < D; ++i) A[i + j * D] = e[i] * ::fma(t2, mr.N2[i + j * D], ::fma(tau, mr.N1[i + j * D], i == j ? 1.0 : 0.0));
        if (first) {      // (the series' first step only: one lane of one wave, once)
            TGP_UNROLL for (int i = 0; i < D * D; ++i) A[i] = mc.A[i];
        }
    } else {
        TGP_UNROLL for (int i = 0; i < D * D; ++i) A[i] = mr.A[i];
    }
}

// predict (lgc.jl:46-52) on the register-resident model.  AP_out (optional, full d x d): A Symmetric(P) of the INCOMING covariance -- what
// invert_dynamics multiplies again (lgssm.jl:234).
template <int D, bool SDE> TGP_HD void predict_r(const ModelR<D, SDE>& mr, const double* A, double* m, double* P, double* AP_out = nullptr) {
    double mp[D];
    TGP_UNROLL for (int i = 0; i < D; ++i) {
        double acc = mr.a[i];
        TGP_UNROLL for (int k = 0; k < D; ++k) acc = ::fma(A[i + k * D], m[k], acc);
        mp[i] = acc;
    }
    TGP_UNROLL for (int i = 0; i < D; ++i) m[i] = mp[i];
    double AX[D * D], Pn[SD<D>::DS];
    if constexpr (SDE) {
        if (AP_out) {      // A P and A Pinf separately: the smoother needs A P itself
            double AG[D * D];
            mul_A_sym<D>(A, P, AP_out);
            mul_A_sym<D>(A, mr.Qg, AG);
            TGP_UNROLL for (int i = 0; i < D * D; ++i) AX[i] = AP_out[i] - AG[i];
        } else {
            double Dm[SD<D>::DS];
            TGP_UNROLL for (int i = 0; i < SD<D>::DS; ++i) Dm[i] = P[i] - mr.Qg[i];
            mul_A_sym<D>(A, Dm, AX);
        }
    } else {
        mul_A_sym<D>(A, P, AX);
        if (AP_out) { TGP_UNROLL for (int i = 0; i < D * D; ++i) AP_out[i] = AX[i]; }
    }
    mul_AXAt_plus<D>(AX, A, mr.Qg, Pn);
    TGP_UNROLL for (int i = 0; i < SD<D>::DS; ++i) P[i] = Pn[i];
}

// One step of the reverse-time recursion on the register-resident model (see smooth_step above: the same arithmetic)
template <int D, bool SDE> TGP_HD void smooth_step_r(const ModelR<D, SDE>& mr, const double* A, const State<D>& xf, State<D>& xs, bool& ok) {
    constexpr int DS = SD<D>::DS;
    double mp[D], Pp[DS], AP[D * D];
    TGP_UNROLL for (int i = 0; i < D; ++i) mp[i] = xf.m[i];
    TGP_UNROLL for (int i = 0; i < DS; ++i) Pp[i] = xf.P[i];
    predict_r<D, SDE>(mr, A, mp, Pp, AP);
    double U[D * D], inv[D];
    TGP_UNROLL for (int j = 0; j < D; ++j) {
        TGP_UNROLL for (int i = 0; i < j; ++i) {
            double acc = Pp[pidx(i, j)];
            TGP_UNROLL for (int k = 0; k < i; ++k) acc = ::fma(-U[k + i * D], U[k + j * D], acc);
            U[i + j * D] = acc * inv[i];
        }
        double acc = Pp[pidx(j, j)] + kJitter;
        TGP_UNROLL for (int k = 0; k < j; ++k) acc = ::fma(-U[k + j * D], U[k + j * D], acc);
        ok = ok && (acc > 0.0);
        double s, rs;
        fast_sqrt_rsqrt(acc, s, rs);
        U[j + j * D] = s;
        inv[j] = rs;
    }
    double W[D * D];
    TGP_UNROLL for (int j = 0; j < D; ++j) {
        TGP_UNROLL for (int i = 0; i < D; ++i) {
            double acc = AP[i + j * D];
            TGP_UNROLL for (int k = 0; k < i; ++k) acc = ::fma(-U[k + i * D], W[k + j * D], acc);
            W[i + j * D] = acc * inv[i];
        }
        TGP_UNROLL for (int i = D - 1; i >= 0; --i) {
            double acc = W[i + j * D];
            TGP_UNROLL for (int k = i + 1; k < D; ++k) acc = ::fma(-U[i + k * D], W[k + j * D], acc);
            W[i + j * D] = acc * inv[i];
        }
    }
    double dm[D], Dm[DS];
    TGP_UNROLL for (int i = 0; i < D; ++i) dm[i] = xs.m[i] - mp[i];
    TGP_UNROLL for (int j = 0; j < D; ++j)
        TGP_UNROLL for (int i = 0; i <= j; ++i) Dm[pidx(i, j)] = xs.P[pidx(i, j)] - Pp[pidx(i, j)] - (i == j ? kJitter : 0.0);
    TGP_UNROLL for (int i = 0; i < D; ++i) {
        double acc = xf.m[i];
        TGP_UNROLL for (int k = 0; k < D; ++k) acc = ::fma(W[k + i * D], dm[k], acc);
        xs.m[i] = acc;
    }
    double GD[D * D];
    TGP_UNROLL for (int j = 0; j < D; ++j)
        TGP_UNROLL for (int i = 0; i < D; ++i) {
            double acc = 0.0;
            TGP_UNROLL for (int k = 0; k < D; ++k) acc = ::fma(W[k + i * D], Dm[pidx(k, j)], acc);
            GD[i + j * D] = acc;
        }
    TGP_UNROLL for (int j = 0; j < D; ++j)
        TGP_UNROLL for (int i = 0; i <= j; ++i) {
            double acc = xf.P[pidx(i, j)];
            TGP_UNROLL for (int k = 0; k < D; ++k) acc = ::fma(GD[i + k * D], W[k + j * D], acc);
            xs.P[pidx(i, j)] = acc;
        }
}

// The reverse-time recursion over a window of steps [t0, te), composed FORWARDS (round 5, later): xs_(t0) = E xs_(te-1) + g,
// Ps_(t0) = E Ps_(te-1) E' + L.  The forward run has, at step t, everything step t - 1's reverse-time element needs (the filtering state in
// front of the predict, A P, the predicted state: invert_dynamics lgssm.jl:231-238), so the smoothing state a chunk hands to the lane before
// it -- the backward warm-up's result -- comes out of the forward run of the chunk's first Wb steps at ~190 instructions per step, instead
// of a backward pass of its own over them (recompute + reverse step: ~330).
template <int D> struct RevAcc {
    double E[D * D], g[D], L[SD<D>::DS];
    TGP_HD void reset() {
        TGP_UNROLL for (int j = 0; j < D; ++j) TGP_UNROLL for (int i = 0; i < D; ++i) E[i + j * D] = i == j ? 1.0 : 0.0;
        TGP_UNROLL for (int i = 0; i < D; ++i) g[i] = 0.0;
        TGP_UNROLL for (int i = 0; i < SD<D>::DS; ++i) L[i] = 0.0;
    }
    // xs_(t0) from the window's last filtering state
    TGP_HD void finish(const State<D>& xl, State<D>& out) const {
        TGP_UNROLL for (int i = 0; i < D; ++i) {
            double acc = g[i];
            TGP_UNROLL for (int k = 0; k < D; ++k) acc = ::fma(E[i + k * D], xl.m[k], acc);
            out.m[i] = acc;
        }
        double EP[D * D];
        mul_A_sym<D>(E, xl.P, EP);
        mul_AXAt_plus<D>(EP, E, L, out.P);
    }
};
// xf: filtering state of step t - 1; (mp, Pp): predicted state of step t; AP = A Symmetric(xf.P).  Appends step t - 1's element.
template <int D> TGP_HD void rev_append(const State<D>& xf, const double* mp, const double* Pp, const double* AP, RevAcc<D>& ra, bool& ok) {
    constexpr int DS = SD<D>::DS;
    double U[D * D], inv[D];
    TGP_UNROLL for (int j = 0; j < D; ++j) {
        TGP_UNROLL for (int i = 0; i < j; ++i) {
            double acc = Pp[pidx(i, j)];
            TGP_UNROLL for (int k = 0; k < i; ++k) acc = ::fma(-U[k + i * D], U[k + j * D], acc);
            U[i + j * D] = acc * inv[i];
        }
        double acc = Pp[pidx(j, j)] + kJitter;
        TGP_UNROLL for (int k = 0; k < j; ++k) acc = ::fma(-U[k + j * D], U[k + j * D], acc);
        ok = ok && (acc > 0.0);
        double sq, rs;
        fast_sqrt_rsqrt(acc, sq, rs);
        U[j + j * D] = sq;
        inv[j] = rs;
    }
    double W[D * D], Z[D * D];      // Z = U^-T (A Pf) = U G';  W = U^-1 Z = G'
    TGP_UNROLL for (int j = 0; j < D; ++j) {
        TGP_UNROLL for (int i = 0; i < D; ++i) {
            double acc = AP[i + j * D];
            TGP_UNROLL for (int k = 0; k < i; ++k) acc = ::fma(-U[k + i * D], Z[k + j * D], acc);
            Z[i + j * D] = acc * inv[i];
        }
        TGP_UNROLL for (int i = D - 1; i >= 0; --i) {
            double acc = Z[i + j * D];
            TGP_UNROLL for (int k = i + 1; k < D; ++k) acc = ::fma(-U[i + k * D], W[k + j * D], acc);
            W[i + j * D] = acc * inv[i];
        }
    }
    // g_t = mf - G mp;  L_t = Pf - (U G')'(U G')   (lgssm.jl:236-237)
    double gt[D], Lt[DS];
    TGP_UNROLL for (int i = 0; i < D; ++i) {
        double acc = xf.m[i];
        TGP_UNROLL for (int k = 0; k < D; ++k) acc = ::fma(-W[k + i * D], mp[k], acc);
        gt[i] = acc;
    }
    TGP_UNROLL for (int j = 0; j < D; ++j)
        TGP_UNROLL for (int i = 0; i <= j; ++i) {
            double acc = xf.P[pidx(i, j)];
            TGP_UNROLL for (int k = 0; k < D; ++k) acc = ::fma(-Z[k + i * D], Z[k + j * D], acc);
            Lt[pidx(i, j)] = acc;
        }
    // g += E g_t;  L += E L_t E';  E <- E G      (G[i][k] = W[k + i D])
    TGP_UNROLL for (int i = 0; i < D; ++i) {
        double acc = ra.g[i];
        TGP_UNROLL for (int k = 0; k < D; ++k) acc = ::fma(ra.E[i + k * D], gt[k], acc);
        ra.g[i] = acc;
    }
    double EL[D * D], Ln[DS], En[D * D];
    mul_A_sym<D>(ra.E, Lt, EL);
    mul_AXAt_plus<D>(EL, ra.E, ra.L, Ln);
    TGP_UNROLL for (int i = 0; i < DS; ++i) ra.L[i] = Ln[i];
    TGP_UNROLL for (int j = 0; j < D; ++j)
        TGP_UNROLL for (int i = 0; i < D; ++i) {
            double acc = 0.0;
            TGP_UNROLL for (int k = 0; k < D; ++k) acc = ::fma(ra.E[i + k * D], W[j + k * D], acc);      // (E G)[i][j] = sum_k E[i][k] G[k][j], G[k][j] = W[j + k D]
            En[i + j * D] = acc;
        }
    TGP_UNROLL for (int i = 0; i < D * D; ++i) ra.E[i] = En[i];
}

// Inputs of the B steps of a block.  XS bit 0: the noise variance is per step, bit 1: the emission offset is.  A lane's B steps are B
// consecutive values of each stream: the loads of a block are issued together (one cache line per lane and stream: the first load brings
// it, the others hit -- one load per step, spread over the block, was measured 25 % slower: the line is gone from the 32 KB L1 by then),
// a whole block ahead of their use (cur / nxt).
template <int D, bool SDE, int XS, int B> struct Inputs {
    double y[B];
    double R[(XS & 1) ? B : 1];
    double hh[(XS & 2) ? B : 1];
    double tau[SDE ? B : 1];
    unsigned mk[B];      // the steps' mask bytes as loaded (compared where a step uses them: a comparison at the load would wait for it)
    TGP_HD bool obs(int j, long long t, long long T) const { return t < T && mk[j] == 0u; }      // (steps behind the series are missing: the runs pad their last block)
};
TGP_HD long long clamp_t(long long t, long long T) { return t < 0 ? 0 : (t >= T ? T - 1 : t); }

template <int D, bool SDE, int XS, int B> TGP_HD void load_inputs(const KArgs<D>& ka, long long tb, Inputs<D, SDE, XS, B>& in) {
    long long tc[B];
    TGP_UNROLL for (int j = 0; j < B; ++j) tc[j] = clamp_t(tb + j, ka.T);
    TGP_UNROLL for (int j = 0; j < B; ++j) in.y[j] = ka.st.y[tc[j]];
    if constexpr (SDE) { TGP_UNROLL for (int j = 0; j < B; ++j) in.tau[j] = ka.st.tau[tc[j]]; }
    if constexpr ((XS & 1) != 0) { TGP_UNROLL for (int j = 0; j < B; ++j) in.R[j] = ka.st.R[tc[j]]; }
    if constexpr ((XS & 2) != 0) { TGP_UNROLL for (int j = 0; j < B; ++j) in.hh[j] = ka.st.hh[tc[j]]; }
    TGP_UNROLL for (int j = 0; j < B; ++j) in.mk[j] = 0u;
    if (ka.st.mask != nullptr) { TGP_UNROLL for (int j = 0; j < B; ++j) in.mk[j] = ka.st.mask[tc[j]]; }
}

// One forward run of a lane: `nblk` blocks of B steps from step ts on (ts a multiple of 8, possibly negative); the blocks that start in
// [lo, hi) are processed, whole (lo, hi multiples of 8; steps behind the series' end are missing).  x: filtering state in front of step
// max(ts, lo) on entry, behind the last processed step on return.  acc: the log marginal likelihood's sums (want_lml: the logs are taken).
// ckpt (null: none): the state in front of every block is kept, [block][component][lane].
// REV: the reverse-time recursion over the steps (rev_t0, rev_te) is composed on the way (see RevAcc: rev holds the composition so far, the run may
// be one of several over the window); *rev_out = the smoothing state of step rev_t0 it gives from the filtering state of step rev_te - 1.
template <int D, bool SDE, int XS, int B, bool REV = false>
TGP_HD void forward_run(const KArgs<D>& ka, const ModelR<D, SDE>& mr, long long ts, int nblk, long long lo, long long hi, State<D>& x, LmlAcc& acc,
                        bool want_lml, double* ckpt, int lane, bool& ok, RevAcc<D>* rev = nullptr, long long rev_t0 = 0, long long rev_te = 0,
                        State<D>* rev_out = nullptr) {
    constexpr int NS = SD<D>::NS;
    Inputs<D, SDE, XS, B> in, nx;
    load_inputs<D, SDE, XS, B>(ka, ts, in);
    for (int b = 0; b < nblk; ++b) {
        const long long tb = ts + (long long)b * B;
        load_inputs<D, SDE, XS, B>(ka, tb + B, nx);      // in flight while this block computes (behind the last block: clamped, unused)
        TGP_ISSUE_BARRIER();
        if (ckpt != nullptr) {
            double* q = ckpt + (size_t)b * NS * 64 + lane;
            TGP_UNROLL for (int k = 0; k < D; ++k) q[(size_t)k * 64] = x.m[k];
            TGP_UNROLL for (int k = 0; k < SD<D>::DS; ++k) q[(size_t)(D + k) * 64] = x.P[k];
        }
        if (tb >= lo && tb < hi) {
            TGP_UNROLL for (int j = 0; j < B; ++j) {
                double A[D * D];
                step_A<D, SDE>(ka.mc, mr, SDE ? in.tau[j] : 0.0, SDE && tb + j == 0, A);
                if constexpr (REV) {
                    const long long t = tb + j;
                    State<D> xf = x;
                    double AP[D * D];
                    predict_r<D, SDE>(mr, A, x.m, x.P, AP);
                    if (t > rev_t0 && t < rev_te) rev_append<D>(xf, x.m, x.P, AP, *rev, ok);
                    update<D>(mr.H, (XS & 2) ? in.hh[j] : mr.hh, (XS & 1) ? in.R[j] : mr.R, in.y[j], in.obs(j, t, ka.T), x.m, x.P, &acc, ok);
                    if (t == rev_te - 1) rev->finish(x, *rev_out);
                } else {
                    predict_r<D, SDE>(mr, A, x.m, x.P);
                    update<D>(mr.H, (XS & 2) ? in.hh[j] : mr.hh, (XS & 1) ? in.R[j] : mr.R, in.y[j], in.obs(j, tb + j, ka.T), x.m, x.P, &acc, ok);
                }
            }
        }
        if (want_lml) acc.flush();      // (wave-uniform)
        else acc.prod = 1.0;
        in = nx;
    }
}

// One backward run of a lane over the blocks nblk - 1 .. 0 of its chunk (first step t0, a multiple of 8): the steps below hi are smoothed.
// fresh: the run starts at step hi - 1 from that step's filtering state (the end of the series, or a warm-up); otherwise xs holds the smoothing
// state of step hi (the next lane's first step).  On return xs is the smoothing state of step t0.  emit_out: write mean / var of the steps.
// The filtering states of a block are recomputed from its checkpoint into LDS (sF: [step][component][lane]) -- with the steps behind the
// series' end as missing ones, exactly as the forward run took them.
template <int D, bool SDE, int XS, int B>
TGP_HD void backward_run(const KArgs<D>& ka, const ModelR<D, SDE>& mr, long long t0, int nblk, long long hi, bool fresh, State<D>& xs, bool emit_out,
                         const double* ckpt, double* sF, int lane, bool& ok) {
    constexpr int NS = SD<D>::NS, DS = SD<D>::DS;
    const long long T = ka.T;
    Inputs<D, SDE, XS, B> in, nx;
    double tau_next = 0.0;      // gap in front of the step behind the block in hand
    const long long tlast = t0 + (long long)(nblk - 1) * B;
    if (SDE) tau_next = ka.st.tau[clamp_t(tlast + B, T)];
    load_inputs<D, SDE, XS, B>(ka, tlast, in);
    State<D> ck, ckN;
    {
        const double* q = ckpt + (size_t)(nblk - 1) * NS * 64 + lane;
        TGP_UNROLL for (int k = 0; k < D; ++k) ck.m[k] = q[(size_t)k * 64];
        TGP_UNROLL for (int k = 0; k < DS; ++k) ck.P[k] = q[(size_t)(D + k) * 64];
    }
    for (int b = nblk - 1; b >= 0; --b) {
        const long long tb = t0 + (long long)b * B;
        load_inputs<D, SDE, XS, B>(ka, tb - B, nx);      // the next block (block b - 1): in flight through this one
        {
            const double* q = ckpt + (size_t)(b > 0 ? b - 1 : 0) * NS * 64 + lane;
            TGP_UNROLL for (int k = 0; k < D; ++k) ckN.m[k] = q[(size_t)k * 64];
            TGP_UNROLL for (int k = 0; k < DS; ++k) ckN.P[k] = q[(size_t)(D + k) * 64];
        }
        TGP_ISSUE_BARRIER();
        // ---- the block's filtering states, forwards from its checkpoint
        State<D> x = ck;
        TGP_UNROLL for (int j = 0; j < B; ++j) {
            double A[D * D];
            step_A<D, SDE>(ka.mc, mr, SDE ? in.tau[j] : 0.0, SDE && tb + j == 0, A);
            predict_r<D, SDE>(mr, A, x.m, x.P);
            update<D>(mr.H, (XS & 2) ? in.hh[j] : mr.hh, (XS & 1) ? in.R[j] : mr.R, in.y[j], in.obs(j, tb + j, T), x.m, x.P, (LmlAcc*)nullptr, ok);
            TGP_UNROLL for (int k = 0; k < D; ++k) sF[(j * NS + k) * 64 + lane] = x.m[k];
            TGP_UNROLL for (int k = 0; k < DS; ++k) sF[(j * NS + D + k) * 64 + lane] = x.P[k];
        }
        // ---- backwards through the block (the barrier: the states are to be READ BACK from LDS -- without it the optimiser forwards the eight
        // states of the first half to the second through registers, which is the register pressure the LDS buffer is there to remove)
        TGP_ISSUE_BARRIER();
        double om[B], ov[B];
        TGP_UNROLL for (int j = B - 1; j >= 0; --j) {
            const long long t = tb + j;
            om[j] = 0.0;
            ov[j] = 0.0;
            if (t < hi) {
                double rn = 0.0;
                if (emit_out) rn = ka.st.Rnew[ka.st.rnew_per_step ? t : 0];      // (asked for here, used at the end of the step)
                State<D> xf;
                TGP_UNROLL for (int k = 0; k < D; ++k) xf.m[k] = sF[(j * NS + k) * 64 + lane];
                TGP_UNROLL for (int k = 0; k < DS; ++k) xf.P[k] = sF[(j * NS + D + k) * 64 + lane];
                if (fresh && t == hi - 1) {
                    xs = xf;
                } else {
                    double A[D * D];
                    step_A<D, SDE>(ka.mc, mr, SDE ? (j == B - 1 ? tau_next : in.tau[j + 1 < B ? j + 1 : j]) : 0.0, false, A);
                    smooth_step_r<D, SDE>(mr, A, xf, xs, ok);
                }
                if (emit_out) emit<D>(mr.H, (XS & 2) ? in.hh[j] : mr.hh, rn, xs, om[j], ov[j]);
            }
        }
        if (emit_out) {
#if defined(__HIP_DEVICE_COMPILE__)
            // A lane's B outputs are B consecutive values of each array, 8 C bytes from its neighbour's: stored by their owners they are
            // 64 pieces of 8 bytes per instruction (measured: the main backward pass took 2.4 x the warm-up pass, which stores nothing).
            // Through LDS instead (the block's states have been read: their buffer is free): rows of B values, B lanes store one row's
            // B x 8 consecutive bytes.
            constexpr int LD = B + 1, RPI = 64 / B;      // row stride in doubles (bank spread), rows per store instruction
            const unsigned long long emits = __builtin_amdgcn_ballot_w64(hi > t0);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            TGP_UNROLL for (int j = 0; j < B; ++j) {
                sF[lane * LD + j] = om[j];
                sF[64 * LD + lane * LD + j] = ov[j];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // first step of lane 0's chunk, from lane 1's (always a chunk of the series; a lane without one has t0 = 0)
            const long long tw0 = (((long long)__builtin_amdgcn_readlane((int)(t0 >> 32), 1) << 32) | (unsigned)__builtin_amdgcn_readlane((int)t0, 1)) - ka.C;
            const bool wide = (((unsigned long long)ka.mean | (unsigned long long)ka.var) & 15ull) == 0ull;      // (wave-uniform)
            if (wide) {      // 16-byte pieces: B / 2 lanes per row, 128 / B rows per instruction
                constexpr int PPR = B / 2, RPW = 64 / PPR;
                const int pc = lane % PPR, rw = lane / PPR;
                TGP_UNROLL for (int k = 0; k < PPR; ++k) {
                    const int r = rw + RPW * k;
                    const long long t = tw0 + (long long)r * ka.C + (tb - t0) + 2 * pc;
                    double2 vm, vv;
                    vm.x = sF[r * LD + 2 * pc];
                    vm.y = sF[r * LD + 2 * pc + 1];
                    vv.x = sF[64 * LD + r * LD + 2 * pc];
                    vv.y = sF[64 * LD + r * LD + 2 * pc + 1];
                    if (((emits >> r) & 1ull) != 0ull) {
                        if (t + 1 < T) {
                            *reinterpret_cast<double2*>(ka.mean + t) = vm;
                            *reinterpret_cast<double2*>(ka.var + t) = vv;
                        } else if (t < T) {
                            ka.mean[t] = vm.x;
                            ka.var[t] = vv.x;
                        }
                    }
                }
            } else {
                const int e = lane % B, r0 = lane / B;
                TGP_UNROLL for (int k = 0; k < B; ++k) {
                    const int r = r0 + RPI * k;
                    const long long t = tw0 + (long long)r * ka.C + (tb - t0) + e;
                    const double vm = sF[r * LD + e], vv = sF[64 * LD + r * LD + e];
                    if (((emits >> r) & 1ull) != 0ull && t < T) {
                        ka.mean[t] = vm;
                        ka.var[t] = vv;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#else
            TGP_UNROLL for (int j = 0; j < B; ++j) {
                if (tb + j < hi) {
                    ka.mean[tb + j] = om[j];
                    ka.var[tb + j] = ov[j];
                }
            }
#endif
        }
        if (SDE) tau_next = in.tau[0];
        in = nx;
        ck = ckN;
    }
}

}  // namespace tgp_sweep
