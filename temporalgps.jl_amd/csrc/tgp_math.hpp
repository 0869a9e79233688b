// Small fixed-size fp64 linear algebra + the two scan monoids of the time-parallel Kalman engine.
// Everything is a template on the state dimension D with fully unrolled loops, so that on gfx950
// every matrix lives in VGPRs of ONE lane (one scan element / one chunk of steps per lane).
// Column-major d x d blocks throughout (the reference's Julia layout).
//
// Reference semantics restated here (file:line under /root/reference/src):
//   predict              models/linear_gaussian_conditionals.jl:46-52   (A*Symmetric(P))*A' + Q
//   posterior_and_lml    models/linear_gaussian_conditionals.jl:247-257 (ScalarOutputLGC)
//   invert_dynamics      models/lgssm.jl:231-238 (jitter 1e-10)
// The associative-scan elements (filter monoid (Abar,b,C,eta,J), affine monoid (E,g,L)) are NOT in
// the reference (its scan is a sequential loop, util/scan.jl:15-28); they follow Sarkka &
// Garcia-Fernandez, "Temporal Parallelization of Bayesian Smoothers" (IEEE TAC 2021), SURVEY.md 7.3.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TGP_HD __host__ __device__ __forceinline__
#else
#define TGP_HD inline
#endif

#define TGP_UNROLL _Pragma("unroll")

// State dimensions >= TGP_BIG_D do not fit one lane's register file (d = 6: ~800 VGPRs of live matrices
// against 512): fully inlined, hipcc 7.2 spills hundreds of VGPRs and SGPRs and we measured silently wrong
// results from that path. For those D the per-step / per-combine building blocks are real (noinline)
// device functions: each has a small register footprint and the matrices live in the lane's private
// (scratch) memory by construction. d <= 4 stays fully inlined in registers.
#define TGP_BIG_D 5
#if defined(__HIPCC__)
#define TGP_NOINLINE __host__ __device__ __attribute__((noinline))
#else
#define TGP_NOINLINE __attribute__((noinline))
#endif

namespace tgp {

constexpr double kLog2Pi = 1.8378770664093454835606594728112;
constexpr double kLargeVar = 1e15;  // missings.jl:43

template <int D> struct Dim {
    static constexpr int DD = D * D;
    static constexpr int DS = D * (D + 1) / 2;       // packed symmetric
    static constexpr int NS = D + DS;                // state (m, P) packed
    static constexpr int NF = DD + D + DS + D + DS;  // filter element packed
    static constexpr int NA = DD + D + DS;           // affine element packed
};

// ---------------------------------------------------------------- basic products
template <int D> TGP_HD void mat_mul(const double* A, const double* B, double* C) {  // C = A B
    TGP_UNROLL for (int j = 0; j < D; ++j) TGP_UNROLL for (int i = 0; i < D; ++i) {
        double acc = 0.0;
        TGP_UNROLL for (int k = 0; k < D; ++k) acc = fma(A[i + k * D], B[k + j * D], acc);
        C[i + j * D] = acc;
    }
}
template <int D> TGP_HD void mat_mul_nt(const double* A, const double* B, double* C) {  // C = A B'
    TGP_UNROLL for (int j = 0; j < D; ++j) TGP_UNROLL for (int i = 0; i < D; ++i) {
        double acc = 0.0;
        TGP_UNROLL for (int k = 0; k < D; ++k) acc = fma(A[i + k * D], B[j + k * D], acc);
        C[i + j * D] = acc;
    }
}
template <int D> TGP_HD void mat_mul_tn(const double* A, const double* B, double* C) {  // C = A' B
    TGP_UNROLL for (int j = 0; j < D; ++j) TGP_UNROLL for (int i = 0; i < D; ++i) {
        double acc = 0.0;
        TGP_UNROLL for (int k = 0; k < D; ++k) acc = fma(A[k + i * D], B[k + j * D], acc);
        C[i + j * D] = acc;
    }
}
template <int D> TGP_HD void mat_vec(const double* A, const double* x, double* y) {  // y = A x
    TGP_UNROLL for (int i = 0; i < D; ++i) {
        double acc = 0.0;
        TGP_UNROLL for (int k = 0; k < D; ++k) acc = fma(A[i + k * D], x[k], acc);
        y[i] = acc;
    }
}
template <int D> TGP_HD void mat_tvec(const double* A, const double* x, double* y) {  // y = A' x
    TGP_UNROLL for (int i = 0; i < D; ++i) {
        double acc = 0.0;
        TGP_UNROLL for (int k = 0; k < D; ++k) acc = fma(A[k + i * D], x[k], acc);
        y[i] = acc;
    }
}
template <int D> TGP_HD void set_identity(double* A) {
    TGP_UNROLL for (int j = 0; j < D; ++j) TGP_UNROLL for (int i = 0; i < D; ++i) A[i + j * D] = (i == j) ? 1.0 : 0.0;
}
template <int N> TGP_HD void set_zero(double* x) { TGP_UNROLL for (int i = 0; i < N; ++i) x[i] = 0.0; }
template <int N> TGP_HD void copy_n(const double* s, double* d) { TGP_UNROLL for (int i = 0; i < N; ++i) d[i] = s[i]; }

// Symmetric(P): mirror the upper triangle (lgc.jl:50 "needed for numerical stability").
template <int D> TGP_HD void sym_upper(double* P) {
    TGP_UNROLL for (int j = 0; j < D; ++j) TGP_UNROLL for (int i = j + 1; i < D; ++i) P[i + j * D] = P[j + i * D];
}
// (P + P')/2, used only on scan-combine results (not reference arithmetic).
template <int D> TGP_HD void symmetrize(double* P) {
    TGP_UNROLL for (int j = 0; j < D; ++j) TGP_UNROLL for (int i = j + 1; i < D; ++i) {
        double v = 0.5 * (P[i + j * D] + P[j + i * D]);
        P[i + j * D] = v;
        P[j + i * D] = v;
    }
}

// packed symmetric <-> full (upper triangle, column by column)
template <int D, typename Store> TGP_HD void store_sym(const double* P, Store st, int base) {
    int n = base;
    TGP_UNROLL for (int j = 0; j < D; ++j) TGP_UNROLL for (int i = 0; i <= j; ++i) st(n++, P[i + j * D]);
}
template <int D, typename Load> TGP_HD void load_sym(double* P, Load ld, int base) {
    int n = base;
    TGP_UNROLL for (int j = 0; j < D; ++j) TGP_UNROLL for (int i = 0; i <= j; ++i) {
        double v = ld(n++);
        P[i + j * D] = v;
        P[j + i * D] = v;
    }
}

// ---------------------------------------------------------------- reference per-step maths
// predict: m <- A m + a ; P <- (A Symmetric(P)) A' + Q
template <int D> TGP_HD void predict_impl(const double* A, const double* a, const double* Q, double* m, double* P) {
    double mp[D], AS[D * D];
    sym_upper<D>(P);
    mat_vec<D>(A, m, mp);
    TGP_UNROLL for (int i = 0; i < D; ++i) m[i] = mp[i] + a[i];
    mat_mul<D>(A, P, AS);
    TGP_UNROLL for (int j = 0; j < D; ++j) TGP_UNROLL for (int i = 0; i < D; ++i) {
        double acc = 0.0;
        TGP_UNROLL for (int k = 0; k < D; ++k) acc = fma(AS[i + k * D], A[j + k * D], acc);
        P[i + j * D] = acc + Q[i + j * D];
    }
}

// ScalarOutputLGC update; returns lml. `ok` is cleared when S is not positive (Julia: DomainError).
template <int D> TGP_HD double update_scalar_impl(const double* H, double h, double R, double y, double* m, double* P, bool& ok) {
    double V[D];
    TGP_UNROLL for (int j = 0; j < D; ++j) {
        double acc = 0.0;
        TGP_UNROLL for (int k = 0; k < D; ++k) acc = fma(H[k], P[k + j * D], acc);
        V[j] = acc;
    }
    double s2 = 0.0, hm = 0.0;
    TGP_UNROLL for (int k = 0; k < D; ++k) { s2 = fma(V[k], H[k], s2); hm = fma(H[k], m[k], hm); }
    double S = s2 + R;
    ok = ok && (S > 0.0);
    double sqrtS = sqrt(S);
    double inv = 1.0 / sqrtS;
    double alpha = (y - (hm + h)) * inv;
    double B[D];
    TGP_UNROLL for (int j = 0; j < D; ++j) B[j] = V[j] * inv;
    TGP_UNROLL for (int i = 0; i < D; ++i) m[i] = fma(B[i], alpha, m[i]);
    TGP_UNROLL for (int j = 0; j < D; ++j) TGP_UNROLL for (int i = 0; i < D; ++i) P[i + j * D] = fma(-B[i], B[j], P[i + j * D]);
    return -(kLog2Pi + 2.0 * log(sqrtS) + alpha * alpha) * 0.5;
}

// Same update with the transcendental work trimmed for the device hot loop: one reciprocal of S instead of
// sqrt + reciprocal (B'B = V'V / S, B'alpha = V' v / S), and the log is left to the caller, who takes ONE log
// of the product of up to 8 consecutive S (log prod = sum log up to rounding). Returns v^2 / S; S in S_out.
template <int D> TGP_HD double update_scalar_nolog_impl(const double* H, double h, double R, double y, double* m, double* P, bool& ok, double& S_out) {
    double V[D];
    TGP_UNROLL for (int j = 0; j < D; ++j) {
        double acc = 0.0;
        TGP_UNROLL for (int k = 0; k < D; ++k) acc = fma(H[k], P[k + j * D], acc);
        V[j] = acc;
    }
    double s2 = 0.0, hm = 0.0;
    TGP_UNROLL for (int k = 0; k < D; ++k) { s2 = fma(V[k], H[k], s2); hm = fma(H[k], m[k], hm); }
    const double S = s2 + R;
    ok = ok && (S > 0.0);
    const double iS = 1.0 / S;
    const double v = y - (hm + h);
    const double viS = v * iS;
    TGP_UNROLL for (int i = 0; i < D; ++i) m[i] = fma(V[i], viS, m[i]);
    TGP_UNROLL for (int j = 0; j < D; ++j) {
        const double w = V[j] * iS;
        TGP_UNROLL for (int i = 0; i < D; ++i) P[i + j * D] = fma(-V[i], w, P[i + j * D]);
    }
    S_out = S;
    return v * viS;
}

// emission predict for scalar outputs: mean = H'm + h ; var = (H' Symmetric(P)) H + R
template <int D> TGP_HD void emit_scalar(const double* H, double h, double R, const double* m, const double* P, double& mean, double& var) {
    double mu = 0.0, v = 0.0;
    TGP_UNROLL for (int j = 0; j < D; ++j) {
        double acc = 0.0;
        TGP_UNROLL for (int k = 0; k < D; ++k) acc = fma(H[k], (k <= j) ? P[k + j * D] : P[j + k * D], acc);
        v = fma(acc, H[j], v);
        mu = fma(H[j], m[j], mu);
    }
    mean = mu + h;
    var = v + R;
}

// upper Cholesky factor of Symmetric(S) (upper triangle read). Returns false if not PD.
template <int D> TGP_HD bool chol_upper(const double* S, double* U) {
    bool ok = true;
    TGP_UNROLL for (int j = 0; j < D; ++j) {
        TGP_UNROLL for (int i = 0; i <= j; ++i) {
            double acc = S[i + j * D];
            TGP_UNROLL for (int k = 0; k < i; ++k) acc = fma(-U[k + i * D], U[k + j * D], acc);
            if (i == j) {
                ok = ok && (acc > 0.0);
                U[j + j * D] = sqrt(acc);
            } else {
                U[i + j * D] = acc / U[i + i * D];
            }
        }
        TGP_UNROLL for (int i = j + 1; i < D; ++i) U[i + j * D] = 0.0;
    }
    return ok;
}

// invert_dynamics (lgssm.jl:231-238): filtered (mf,Pf), predicted (mp,Pp), transition A -> (G,g,L)
template <int D> TGP_HD bool invert_dynamics_impl(const double* mf, const double* Pf, const double* mp, const double* Pp,
                                             const double* A, double* G, double* g, double* L, double jitter) {
    double Pj[D * D], U[D * D], Gt[D * D], UG[D * D];
    copy_n<D * D>(Pp, Pj);
    TGP_UNROLL for (int i = 0; i < D; ++i) Pj[i + i * D] += jitter;
    bool ok = chol_upper<D>(Pj, U);
    double invd[D];
    TGP_UNROLL for (int i = 0; i < D; ++i) invd[i] = 1.0 / U[i + i * D];
    mat_mul<D>(A, Pf, Gt);  // X = A Pf
    TGP_UNROLL for (int j = 0; j < D; ++j) {
        TGP_UNROLL for (int i = 0; i < D; ++i) {  // U' z = x
            double acc = Gt[i + j * D];
            TGP_UNROLL for (int k = 0; k < i; ++k) acc = fma(-U[k + i * D], Gt[k + j * D], acc);
            Gt[i + j * D] = acc * invd[i];
        }
        TGP_UNROLL for (int i = D - 1; i >= 0; --i) {  // U w = z
            double acc = Gt[i + j * D];
            TGP_UNROLL for (int k = i + 1; k < D; ++k) acc = fma(-U[i + k * D], Gt[k + j * D], acc);
            Gt[i + j * D] = acc * invd[i];
        }
    }
    TGP_UNROLL for (int j = 0; j < D; ++j) TGP_UNROLL for (int i = 0; i < D; ++i) G[i + j * D] = Gt[j + i * D];
    TGP_UNROLL for (int i = 0; i < D; ++i) {
        double acc = 0.0;
        TGP_UNROLL for (int k = 0; k < D; ++k) acc = fma(G[i + k * D], mp[k], acc);
        g[i] = mf[i] - acc;
    }
    TGP_UNROLL for (int j = 0; j < D; ++j) TGP_UNROLL for (int i = 0; i < D; ++i) {
        double acc = 0.0;
        TGP_UNROLL for (int k = i; k < D; ++k) acc = fma(U[i + k * D], Gt[k + j * D], acc);
        UG[i + j * D] = acc;
    }
    TGP_UNROLL for (int j = 0; j < D; ++j) TGP_UNROLL for (int i = 0; i < D; ++i) {
        double acc = 0.0;
        TGP_UNROLL for (int k = 0; k < D; ++k) acc = fma(UG[k + i * D], UG[k + j * D], acc);
        L[i + j * D] = Pf[i + j * D] - acc;
    }
    return ok;
}

// ---------------------------------------------------------------- in-register inverse (partial pivoting, branch-free)
template <int D> TGP_HD void mat_inverse(double* M, double* X) {  // X = M^-1 ; M destroyed
    set_identity<D>(X);
    TGP_UNROLL for (int k = 0; k < D; ++k) {
        int piv = k;
        double best = fabs(M[k + k * D]);
        TGP_UNROLL for (int i = k + 1; i < D; ++i) {
            double v = fabs(M[i + k * D]);
            bool gt = v > best;
            best = gt ? v : best;
            piv = gt ? i : piv;
        }
        TGP_UNROLL for (int i = k + 1; i < D; ++i) {
            bool sw = (piv == i);
            TGP_UNROLL for (int j = 0; j < D; ++j) {
                double t = M[k + j * D], u = M[i + j * D];
                M[k + j * D] = sw ? u : t;
                M[i + j * D] = sw ? t : u;
                double t2 = X[k + j * D], u2 = X[i + j * D];
                X[k + j * D] = sw ? u2 : t2;
                X[i + j * D] = sw ? t2 : u2;
            }
        }
        double inv = 1.0 / M[k + k * D];
        TGP_UNROLL for (int i = k + 1; i < D; ++i) {
            double f = M[i + k * D] * inv;
            TGP_UNROLL for (int j = k + 1; j < D; ++j) M[i + j * D] = fma(-f, M[k + j * D], M[i + j * D]);
            TGP_UNROLL for (int j = 0; j < D; ++j) X[i + j * D] = fma(-f, X[k + j * D], X[i + j * D]);
        }
    }
    TGP_UNROLL for (int k = D - 1; k >= 0; --k) {
        double inv = 1.0 / M[k + k * D];
        TGP_UNROLL for (int j = 0; j < D; ++j) {
            double acc = X[k + j * D];
            TGP_UNROLL for (int i = k + 1; i < D; ++i) acc = fma(-M[k + i * D], X[i + j * D], acc);
            X[k + j * D] = acc * inv;
        }
    }
}

// ---------------------------------------------------------------- state (m, P)
template <int D> struct State {
    double m[D];
    double P[D * D];
};

// ---------------------------------------------------------------- filter monoid
// Element for a run of steps (s, e]:  p(x_e | x_s, y_{s+1:e}) = N(A x_s + b, C),
//                                     p(y_{s+1:e} | x_s)     ∝ exp(eta' x_s - x_s' J x_s / 2).
template <int D> struct FElem {
    double A[D * D], b[D], C[D * D], eta[D], J[D * D];
    TGP_HD void identity() {
        set_identity<D>(A);
        set_zero<D>(b);
        set_zero<D * D>(C);
        set_zero<D>(eta);
        set_zero<D * D>(J);
    }
};

// out = later(j) o earlier(i)
template <int D> TGP_HD void f_combine_impl(const FElem<D>& ei, const FElem<D>& ej, FElem<D>& out) {
    double M[D * D], Minv[D * D], T1[D * D], T2[D * D], u[D], w[D];
    mat_mul<D>(ei.C, ej.J, M);  // C_i J_j
    TGP_UNROLL for (int i = 0; i < D; ++i) M[i + i * D] += 1.0;
    mat_inverse<D>(M, Minv);                 // (I + C_i J_j)^-1
    mat_mul<D>(ej.A, Minv, T1);              // A_j M
    // b = A_j M (b_i + C_i eta_j) + b_j
    mat_vec<D>(ei.C, ej.eta, u);
    TGP_UNROLL for (int i = 0; i < D; ++i) u[i] += ei.b[i];
    mat_vec<D>(T1, u, w);
    // eta = A_i' M' (eta_j - J_j b_i) + eta_i ;  J = A_i' M' J_j A_i + J_i
    double v[D], z[D];
    mat_vec<D>(ej.J, ei.b, v);
    TGP_UNROLL for (int i = 0; i < D; ++i) v[i] = ej.eta[i] - v[i];
    mat_tvec<D>(Minv, v, z);                 // M' v
    double neta[D];
    mat_tvec<D>(ei.A, z, neta);
    double JA[D * D], MJA[D * D], nJ[D * D];
    mat_mul<D>(ej.J, ei.A, JA);              // J_j A_i
    mat_mul_tn<D>(Minv, JA, MJA);            // M' J_j A_i
    mat_mul_tn<D>(ei.A, MJA, nJ);            // A_i' M' J_j A_i
    // A = A_j M A_i ; C = A_j M C_i A_j' + C_j
    double nA[D * D], nC[D * D];
    mat_mul<D>(T1, ei.A, nA);
    mat_mul<D>(T1, ei.C, T2);
    mat_mul_nt<D>(T2, ej.A, nC);
    TGP_UNROLL for (int i = 0; i < D; ++i) {
        out.b[i] = w[i] + ej.b[i];
        out.eta[i] = neta[i] + ei.eta[i];
    }
    TGP_UNROLL for (int i = 0; i < D * D; ++i) {
        out.A[i] = nA[i];
        out.C[i] = nC[i] + ej.C[i];
        out.J[i] = nJ[i] + ei.J[i];
    }
    symmetrize<D>(out.C);
    symmetrize<D>(out.J);
}

// posterior state after the run, given the state before it
template <int D> TGP_HD void f_apply_impl(const FElem<D>& e, const State<D>& in, State<D>& out) {
    double M[D * D], Minv[D * D], T1[D * D], T2[D * D], u[D], w[D];
    mat_mul<D>(in.P, e.J, M);
    TGP_UNROLL for (int i = 0; i < D; ++i) M[i + i * D] += 1.0;
    mat_inverse<D>(M, Minv);
    mat_mul<D>(e.A, Minv, T1);
    mat_vec<D>(in.P, e.eta, u);
    TGP_UNROLL for (int i = 0; i < D; ++i) u[i] += in.m[i];
    mat_vec<D>(T1, u, w);
    mat_mul<D>(T1, in.P, T2);
    double nP[D * D];
    mat_mul_nt<D>(T2, e.A, nP);
    TGP_UNROLL for (int i = 0; i < D; ++i) out.m[i] = w[i] + e.b[i];
    TGP_UNROLL for (int i = 0; i < D * D; ++i) out.P[i] = nP[i] + e.C[i];
    symmetrize<D>(out.P);
}

// Extend a running element by ONE scalar-output Kalman step (cheap: no solve, rank-one update).
// do_predict == false skips the transition (first processed step of a Reverse-ordered model).
template <int D>
TGP_HD void f_extend_impl(FElem<D>& e, bool do_predict, const double* A, const double* a, const double* Q, const double* H,
                     double h, double R, double y) {
    if (do_predict) {
        double T1[D * D], T2[D * D], bp[D];
        mat_mul<D>(A, e.A, T1);
        copy_n<D * D>(T1, e.A);
        mat_vec<D>(A, e.b, bp);
        TGP_UNROLL for (int i = 0; i < D; ++i) e.b[i] = bp[i] + a[i];
        mat_mul<D>(A, e.C, T1);
        mat_mul_nt<D>(T1, A, T2);
        TGP_UNROLL for (int i = 0; i < D * D; ++i) e.C[i] = T2[i] + Q[i];
    }
    double w[D], Cv[D];
    mat_tvec<D>(e.A, H, w);   // (H' Abar)'
    mat_vec<D>(e.C, H, Cv);
    double s = R, r = y - h;
    TGP_UNROLL for (int k = 0; k < D; ++k) {
        s = fma(H[k], Cv[k], s);
        r = fma(-H[k], e.b[k], r);
    }
    double is = 1.0 / s;
    TGP_UNROLL for (int i = 0; i < D; ++i) {
        e.eta[i] = fma(w[i], r * is, e.eta[i]);
        e.b[i] = fma(Cv[i], r * is, e.b[i]);
    }
    TGP_UNROLL for (int j = 0; j < D; ++j) TGP_UNROLL for (int i = 0; i < D; ++i) {
        e.J[i + j * D] = fma(w[i] * is, w[j], e.J[i + j * D]);
        e.A[i + j * D] = fma(-Cv[i] * is, w[j], e.A[i + j * D]);
        e.C[i + j * D] = fma(-Cv[i] * is, Cv[j], e.C[i + j * D]);
    }
}

// ---------------------------------------------------------------- affine monoid  x' = E x + g + N(0, L)
template <int D> struct AElem {
    double E[D * D], g[D], L[D * D];
    TGP_HD void identity() {
        set_identity<D>(E);
        set_zero<D>(g);
        set_zero<D * D>(L);
    }
};

// out = later(j) o earlier(i)  (processing order; for the smoother "earlier" means later in time)
template <int D, bool COV> TGP_HD void a_combine_impl(const AElem<D>& ei, const AElem<D>& ej, AElem<D>& out) {
    double nE[D * D], ng[D];
    mat_mul<D>(ej.E, ei.E, nE);
    mat_vec<D>(ej.E, ei.g, ng);
    if (COV) {
        double T1[D * D], nL[D * D];
        mat_mul<D>(ej.E, ei.L, T1);
        mat_mul_nt<D>(T1, ej.E, nL);
        TGP_UNROLL for (int i = 0; i < D * D; ++i) out.L[i] = nL[i] + ej.L[i];
        symmetrize<D>(out.L);
    } else {
        set_zero<D * D>(out.L);
    }
    TGP_UNROLL for (int i = 0; i < D; ++i) out.g[i] = ng[i] + ej.g[i];
    copy_n<D * D>(nE, out.E);
}

template <int D, bool COV> TGP_HD void a_apply_impl(const AElem<D>& e, const State<D>& in, State<D>& out) {
    double nm[D];
    mat_vec<D>(e.E, in.m, nm);
    TGP_UNROLL for (int i = 0; i < D; ++i) out.m[i] = nm[i] + e.g[i];
    if (COV) {
        double T1[D * D], nP[D * D];
        mat_mul<D>(e.E, in.P, T1);
        mat_mul_nt<D>(T1, e.E, nP);
        TGP_UNROLL for (int i = 0; i < D * D; ++i) out.P[i] = nP[i] + e.L[i];
        symmetrize<D>(out.P);
    } else {
        set_zero<D * D>(out.P);
    }
}

// running composition in processing order:  e <- step o e   with step = (A, c, Q)
template <int D, bool COV> TGP_HD void a_extend_impl(AElem<D>& e, const double* A, const double* c, const double* Q) {
    double T1[D * D], gp[D];
    mat_mul<D>(A, e.E, T1);
    copy_n<D * D>(T1, e.E);
    mat_vec<D>(A, e.g, gp);
    TGP_UNROLL for (int i = 0; i < D; ++i) e.g[i] = gp[i] + c[i];
    if (COV) {
        double T2[D * D];
        mat_mul<D>(A, e.L, T1);
        mat_mul_nt<D>(T1, A, T2);
        TGP_UNROLL for (int i = 0; i < D * D; ++i) e.L[i] = T2[i] + Q[i];
    }
}

// running composition for the time-REVERSED chain while sweeping forward in time:
//   x_s = E x_{k-1} + ghat + N(0, Lhat),  x_{k-1} = G x_k + g + N(0, L)   =>   e <- e o (G, g, L)
template <int D> TGP_HD void a_extend_right_impl(AElem<D>& e, const double* G, const double* g, const double* L) {
    double T1[D * D], T2[D * D], eg[D];
    mat_vec<D>(e.E, g, eg);
    TGP_UNROLL for (int i = 0; i < D; ++i) e.g[i] += eg[i];
    mat_mul<D>(e.E, L, T1);
    mat_mul_nt<D>(T1, e.E, T2);
    TGP_UNROLL for (int i = 0; i < D * D; ++i) e.L[i] += T2[i];
    mat_mul<D>(e.E, G, T1);
    copy_n<D * D>(T1, e.E);
}

// ---------------------------------------------------------------- inline (d <= 4) / out-of-line (d >= 5) dispatch

template <int D> TGP_NOINLINE void predict_out(const double* A, const double* a, const double* Q, double* m, double* P) { predict_impl<D>(A, a, Q, m, P); }
template <int D> TGP_HD void predict(const double* A, const double* a, const double* Q, double* m, double* P) {
    if constexpr (D >= TGP_BIG_D) { predict_out<D>(A, a, Q, m, P); } else { predict_impl<D>(A, a, Q, m, P); }
}

template <int D> TGP_NOINLINE double update_scalar_out(const double* H, double h, double R, double y, double* m, double* P, bool& ok) { return update_scalar_impl<D>(H, h, R, y, m, P, ok); }
template <int D> TGP_HD double update_scalar(const double* H, double h, double R, double y, double* m, double* P, bool& ok) {
    if constexpr (D >= TGP_BIG_D) { return update_scalar_out<D>(H, h, R, y, m, P, ok); } else { return update_scalar_impl<D>(H, h, R, y, m, P, ok); }
}

template <int D> TGP_NOINLINE double update_scalar_nolog_out(const double* H, double h, double R, double y, double* m, double* P, bool& ok, double& S_out) { return update_scalar_nolog_impl<D>(H, h, R, y, m, P, ok, S_out); }
template <int D> TGP_HD double update_scalar_nolog(const double* H, double h, double R, double y, double* m, double* P, bool& ok, double& S_out) {
    if constexpr (D >= TGP_BIG_D) { return update_scalar_nolog_out<D>(H, h, R, y, m, P, ok, S_out); } else { return update_scalar_nolog_impl<D>(H, h, R, y, m, P, ok, S_out); }
}

template <int D> TGP_NOINLINE bool invert_dynamics_out(const double* mf, const double* Pf, const double* mp, const double* Pp, const double* A, double* G, double* g, double* L, double jitter) { return invert_dynamics_impl<D>(mf, Pf, mp, Pp, A, G, g, L, jitter); }
template <int D> TGP_HD bool invert_dynamics(const double* mf, const double* Pf, const double* mp, const double* Pp, const double* A, double* G, double* g, double* L, double jitter = 1e-10) {
    if constexpr (D >= TGP_BIG_D) { return invert_dynamics_out<D>(mf, Pf, mp, Pp, A, G, g, L, jitter); } else { return invert_dynamics_impl<D>(mf, Pf, mp, Pp, A, G, g, L, jitter); }
}

template <int D> TGP_NOINLINE void f_combine_out(const FElem<D>& ei, const FElem<D>& ej, FElem<D>& out) { f_combine_impl<D>(ei, ej, out); }
template <int D> TGP_HD void f_combine(const FElem<D>& ei, const FElem<D>& ej, FElem<D>& out) {
    if constexpr (D >= TGP_BIG_D) { f_combine_out<D>(ei, ej, out); } else { f_combine_impl<D>(ei, ej, out); }
}

template <int D> TGP_NOINLINE void f_apply_out(const FElem<D>& e, const State<D>& in, State<D>& out) { f_apply_impl<D>(e, in, out); }
template <int D> TGP_HD void f_apply(const FElem<D>& e, const State<D>& in, State<D>& out) {
    if constexpr (D >= TGP_BIG_D) { f_apply_out<D>(e, in, out); } else { f_apply_impl<D>(e, in, out); }
}

template <int D, bool COV> TGP_NOINLINE void a_combine_out(const AElem<D>& ei, const AElem<D>& ej, AElem<D>& out) { a_combine_impl<D, COV>(ei, ej, out); }
template <int D, bool COV> TGP_HD void a_combine(const AElem<D>& ei, const AElem<D>& ej, AElem<D>& out) {
    if constexpr (D >= TGP_BIG_D) { a_combine_out<D, COV>(ei, ej, out); } else { a_combine_impl<D, COV>(ei, ej, out); }
}

template <int D, bool COV> TGP_NOINLINE void a_apply_out(const AElem<D>& e, const State<D>& in, State<D>& out) { a_apply_impl<D, COV>(e, in, out); }
template <int D, bool COV> TGP_HD void a_apply(const AElem<D>& e, const State<D>& in, State<D>& out) {
    if constexpr (D >= TGP_BIG_D) { a_apply_out<D, COV>(e, in, out); } else { a_apply_impl<D, COV>(e, in, out); }
}

template <int D, bool COV> TGP_NOINLINE void a_extend_out(AElem<D>& e, const double* A, const double* c, const double* Q) { a_extend_impl<D, COV>(e, A, c, Q); }
template <int D, bool COV> TGP_HD void a_extend(AElem<D>& e, const double* A, const double* c, const double* Q) {
    if constexpr (D >= TGP_BIG_D) { a_extend_out<D, COV>(e, A, c, Q); } else { a_extend_impl<D, COV>(e, A, c, Q); }
}

template <int D> TGP_NOINLINE void a_extend_right_out(AElem<D>& e, const double* G, const double* g, const double* L) { a_extend_right_impl<D>(e, G, g, L); }
template <int D> TGP_HD void a_extend_right(AElem<D>& e, const double* G, const double* g, const double* L) {
    if constexpr (D >= TGP_BIG_D) { a_extend_right_out<D>(e, G, g, L); } else { a_extend_right_impl<D>(e, G, g, L); }
}

template <int D> TGP_NOINLINE void f_extend_out(FElem<D>& e, bool do_predict, const double* A, const double* a, const double* Q, const double* H, double h, double R, double y) {
    f_extend_impl<D>(e, do_predict, A, a, Q, H, h, R, y);
}
template <int D> TGP_HD void f_extend(FElem<D>& e, bool do_predict, const double* A, const double* a, const double* Q, const double* H, double h, double R, double y) {
    if constexpr (D >= TGP_BIG_D) { f_extend_out<D>(e, do_predict, A, a, Q, H, h, R, y); } else { f_extend_impl<D>(e, do_predict, A, a, Q, H, h, R, y); }
}

// Smoother element of a whole chunk (s, e] WITHOUT per-step work, from quantities the forward pass already
// has: the chunk's filter element, the filtering state before the chunk (xs) and after it (xe).
//   p(x_s | y_{1:e})      = N(mt, Pt),  Pt = (I + P_s J)^-1 P_s,  mt = (I + P_s J)^-1 (m_s + P_s eta)
//   x_e | x_s, y_{s+1:e}  = N(Abar x_s + b, C)            =>  x_s | x_e, y_{1:e} = N(E x_e + g, L)
// which is invert_dynamics applied to (mt, Pt) -> (m_e, P_e) through Abar (no jitter: this is our own
// chunk-level construct; the per-step 1e-10 jitter of lgssm.jl:235 stays where the reference has it).
template <int D> TGP_HD bool chunk_smoother_element(const FElem<D>& e, const State<D>& xs, const State<D>& xe, AElem<D>& r) {
    double M[D * D], Minv[D * D], Pt[D * D], u[D], mt[D];
    mat_mul<D>(xs.P, e.J, M);
    TGP_UNROLL for (int i = 0; i < D; ++i) M[i + i * D] += 1.0;
    mat_inverse<D>(M, Minv);
    mat_mul<D>(Minv, xs.P, Pt);
    symmetrize<D>(Pt);
    mat_vec<D>(xs.P, e.eta, u);
    TGP_UNROLL for (int i = 0; i < D; ++i) u[i] += xs.m[i];
    mat_vec<D>(Minv, u, mt);
    bool ok = invert_dynamics<D>(mt, Pt, xe.m, xe.P, e.A, r.E, r.g, r.L, 0.0);
    symmetrize<D>(r.L);
    return ok;
}

// ---------------------------------------------------------------- packed (SoA) element / state I/O
// `st(k, v)` / `ld(k)` address component k of one element; the kernels supply strided accessors.
template <int D, typename Store> TGP_HD void store_state(const State<D>& s, Store st) {
    TGP_UNROLL for (int i = 0; i < D; ++i) st(i, s.m[i]);
    store_sym<D>(s.P, st, D);
}
template <int D, typename Load> TGP_HD void load_state(State<D>& s, Load ld) {
    TGP_UNROLL for (int i = 0; i < D; ++i) s.m[i] = ld(i);
    load_sym<D>(s.P, ld, D);
}
template <int D, typename Store> TGP_HD void store_felem(const FElem<D>& e, Store st) {
    TGP_UNROLL for (int i = 0; i < D * D; ++i) st(i, e.A[i]);
    TGP_UNROLL for (int i = 0; i < D; ++i) st(D * D + i, e.b[i]);
    store_sym<D>(e.C, st, D * D + D);
    TGP_UNROLL for (int i = 0; i < D; ++i) st(D * D + D + Dim<D>::DS + i, e.eta[i]);
    store_sym<D>(e.J, st, D * D + 2 * D + Dim<D>::DS);
}
template <int D, typename Load> TGP_HD void load_felem(FElem<D>& e, Load ld) {
    TGP_UNROLL for (int i = 0; i < D * D; ++i) e.A[i] = ld(i);
    TGP_UNROLL for (int i = 0; i < D; ++i) e.b[i] = ld(D * D + i);
    load_sym<D>(e.C, ld, D * D + D);
    TGP_UNROLL for (int i = 0; i < D; ++i) e.eta[i] = ld(D * D + D + Dim<D>::DS + i);
    load_sym<D>(e.J, ld, D * D + 2 * D + Dim<D>::DS);
}
template <int D, typename Store> TGP_HD void store_aelem(const AElem<D>& e, Store st) {
    TGP_UNROLL for (int i = 0; i < D * D; ++i) st(i, e.E[i]);
    TGP_UNROLL for (int i = 0; i < D; ++i) st(D * D + i, e.g[i]);
    store_sym<D>(e.L, st, D * D + D);
}
template <int D, typename Load> TGP_HD void load_aelem(AElem<D>& e, Load ld) {
    TGP_UNROLL for (int i = 0; i < D * D; ++i) e.E[i] = ld(i);
    TGP_UNROLL for (int i = 0; i < D; ++i) e.g[i] = ld(D * D + i);
    load_sym<D>(e.L, ld, D * D + D);
}

}  // namespace tgp
