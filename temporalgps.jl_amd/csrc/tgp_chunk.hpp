// Per-lane work of the time-parallel Kalman engine: every lane owns one CHUNK of L0 consecutive
// processing steps and runs it sequentially with all matrices in registers.
//
//   pass 1  chunk_reduce_*   steps of a chunk  -> one scan element        (cheap rank-one extension)
//   (block scans over the chunk elements: tgp_kernels.hpp)
//   pass 2  chunk_apply_*    chunk carry-in state -> the reference's own sequential recursion
//                            (predict / posterior_and_lml / invert_dynamics, bit-for-bit the same
//                            arithmetic as the CPU path once the carry-in is known)
//   pass 3  chunk_smooth     RTS smoother = Reverse step_marginals on the posterior model
//
// "Processing index" r runs 0..T-1 in the order the reference visits steps:
//   Forward (gauss_markov_model.jl:38):  r -> storage index r,      step = predict(r) then emission(r)
//   Reverse (gauss_markov_model.jl:40):  r -> storage index T-1-r,  step = emission(T-1-r) then predict(T-1-r)
// A Reverse model is run as "predict(T-r) [skipped at r = 0] then emission(T-1-r)", which is the same
// sequence of operations (lgssm.jl:111-115, 161-165, 183-187); the trailing predict only feeds the
// returned final state, which no caller on this path reads.
#pragma once
#include "tgp_math.hpp"

namespace tgp {

struct ModelView {
    int64_t T;
    int32_t ordering;  // 0 = Forward, 1 = Reverse
    int32_t pad_;
    const double* A;   // [T|1][d*d] column-major
    const double* a;   // [T|1][d]
    const double* Q;   // [T|1][d*d]
    const double* H;   // [T|1][d]     (ScalarOutputLGC: A = H')
    const double* h;   // [T|1]
    const double* R;   // [T|1]
    int64_t sA, sa, sQ, sH, sh, sR;  // stride in doubles per step; 0 == Fill (shared)
    const double* y;                 // [T]
    const uint8_t* missing;          // [T] or nullptr; 1 => y := 0, R := 1e15 (missings.jl:55-101)
};

TGP_HD int64_t fs_index(int64_t c, int i, int k, int L0, int NS) {
    return ((((c >> 6) * L0 + i) * NS + k) << 6) + (c & 63);
}

// Loads one processing step. LTI == true: A, a, Q, H, h are shared and loaded once (hoisted).
template <int D, bool LTI> struct StepLoader {
    double A[D * D], a[D], Q[D * D], H[D], h, R, y;
    bool do_predict, is_missing;
    int64_t te, tt;

    TGP_HD void init(const ModelView& mv) {
        if (LTI) {
            TGP_UNROLL for (int i = 0; i < D * D; ++i) { A[i] = mv.A[i]; Q[i] = mv.Q[i]; }
            TGP_UNROLL for (int i = 0; i < D; ++i) { a[i] = mv.a[i]; H[i] = mv.H[i]; }
            h = mv.h[0];
        }
    }
    TGP_HD void index(const ModelView& mv, int64_t r) {
        if (mv.ordering == 0) { te = r; tt = r; do_predict = true; }
        else { te = mv.T - 1 - r; tt = mv.T - r; do_predict = (r != 0); }
    }
    TGP_HD void load_transition(const ModelView& mv) {
        if (!LTI && do_predict) {
            const double* pA = mv.A + tt * mv.sA;
            const double* pQ = mv.Q + tt * mv.sQ;
            const double* pa = mv.a + tt * mv.sa;
            TGP_UNROLL for (int i = 0; i < D * D; ++i) { A[i] = pA[i]; Q[i] = pQ[i]; }
            TGP_UNROLL for (int i = 0; i < D; ++i) a[i] = pa[i];
        }
    }
    TGP_HD void load_emission(const ModelView& mv) {
        if (!LTI) {
            const double* pH = mv.H + te * mv.sH;
            TGP_UNROLL for (int i = 0; i < D; ++i) H[i] = pH[i];
            h = mv.h[te * mv.sh];
        }
        R = mv.R[te * mv.sR];
    }
    TGP_HD void load_obs(const ModelView& mv) {
        y = mv.y[te];
        is_missing = (mv.missing != nullptr) && (mv.missing[te] != 0);
        if (is_missing) { y = 0.0; R = kLargeVar; }
    }
    TGP_HD void load(const ModelView& mv, int64_t r) {
        index(mv, r);
        load_transition(mv);
        load_emission(mv);
        load_obs(mv);
    }
};

// ------------------------------------------------------------------------------------------ pass 1
template <int D, bool LTI, typename Store>
TGP_HD void chunk_reduce_filter(const ModelView& mv, int64_t c, int L0, Store st) {
    int64_t r0 = c * (int64_t)L0;
    int64_t r1 = r0 + L0 < mv.T ? r0 + L0 : mv.T;
    FElem<D> e;
    e.identity();
    StepLoader<D, LTI> sl;
    sl.init(mv);
    for (int64_t r = r0; r < r1; ++r) {
        sl.load(mv, r);
        f_extend<D>(e, sl.do_predict, sl.A, sl.a, sl.Q, sl.H, sl.h, sl.R, sl.y);
    }
    store_felem<D>(e, st);
}

// ------------------------------------------------------------------------------------------ pass 2
struct FilterOut {
    double* m_out;  // [T][d]      filtering means      (MODE >= 1, may be null)
    double* P_out;  // [T][d*d]    filtering covariances
    double* fs;     // filtered-state scratch, fs_index layout (MODE 2)
    double* G_out;  // [T][d*d] [T][d] [T][d*d] materialised reverse model (MODE 2, may be null)
    double* g_out;
    double* L_out;
};

struct ChunkStats {
    double lml;
    double nmiss;
    int32_t bad;  // 1 => non-positive innovation variance / Cholesky failure inside this chunk
};

// MODE 0: logpdf only. MODE 1: + filtering distributions. MODE 2: + filtered-state scratch, reverse
// (smoother) chunk element, optional (G, g, L) output.  `rst` stores the reverse element (MODE 2).
template <int D, bool LTI, int MODE, typename RStore>
TGP_HD ChunkStats chunk_apply_filter(const ModelView& mv, int64_t c, int L0, State<D>& x, const FilterOut& out, RStore rst) {
    int64_t r0 = c * (int64_t)L0;
    int64_t r1 = r0 + L0 < mv.T ? r0 + L0 : mv.T;
    ChunkStats cs{0.0, 0.0, 0};
    StepLoader<D, LTI> sl;
    sl.init(mv);
    AElem<D> rev;
    if (MODE == 2) rev.identity();
    bool ok = true;
    for (int64_t r = r0; r < r1; ++r) {
        sl.load(mv, r);
        if (MODE == 2) {
            double mf[D], Pf[D * D];
            copy_n<D>(x.m, mf);
            copy_n<D * D>(x.P, Pf);
            predict<D>(sl.A, sl.a, sl.Q, x.m, x.P);
            double G[D * D], g[D], L[D * D];
            ok = invert_dynamics<D>(mf, Pf, x.m, x.P, sl.A, G, g, L) && ok;
            a_extend_right<D>(rev, G, g, L);
            if (out.G_out) {
                TGP_UNROLL for (int i = 0; i < D * D; ++i) { out.G_out[sl.te * D * D + i] = G[i]; out.L_out[sl.te * D * D + i] = L[i]; }
                TGP_UNROLL for (int i = 0; i < D; ++i) out.g_out[sl.te * D + i] = g[i];
            }
        } else if (sl.do_predict) {
            predict<D>(sl.A, sl.a, sl.Q, x.m, x.P);
        }
        cs.lml += update_scalar<D>(sl.H, sl.h, sl.R, sl.y, x.m, x.P, ok);
        cs.nmiss += sl.is_missing ? 1.0 : 0.0;
        if (MODE >= 1 && out.m_out) {
            TGP_UNROLL for (int i = 0; i < D; ++i) out.m_out[sl.te * D + i] = x.m[i];
        }
        if (MODE >= 1 && out.P_out) {
            TGP_UNROLL for (int i = 0; i < D * D; ++i) out.P_out[sl.te * D * D + i] = x.P[i];
        }
        if (MODE == 2 && out.fs) {
            int i = (int)(r - r0);
            double* fs = out.fs;
            int L0_ = L0;
            store_state<D>(x, [=](int k, double v) { fs[fs_index(c, i, k, L0_, Dim<D>::NS)] = v; });
        }
    }
    if (MODE == 2) store_aelem<D>(rev, rst);
    cs.bad = ok ? 0 : 1;
    return cs;
}

// ------------------------------------------------------------------------------------------ pass 3
// RTS smoother on chunk c. `xs` = smoothed state at the chunk's LAST step; `carry` = filtered state
// just before the chunk's first step. Emits N(H x + h, H P H' + Rnew) for every step (lgssm.jl:111-115
// on the posterior model with replace_observation_noise_cov, missings.jl:35-41).
template <int D, bool LTI>
TGP_HD int chunk_smooth(const ModelView& mv, int64_t c, int L0, State<D>& xs, const State<D>& carry, const double* fs,
                        const double* Rnew, int64_t sRn, double* mean_out, double* var_out) {
    int64_t r0 = c * (int64_t)L0;
    int64_t r1 = r0 + L0 < mv.T ? r0 + L0 : mv.T;
    StepLoader<D, LTI> sl;
    sl.init(mv);
    bool ok = true;
    for (int64_t r = r1 - 1; r >= r0; --r) {
        sl.index(mv, r);
        sl.load_transition(mv);
        sl.load_emission(mv);
        double mean, var;
        emit_scalar<D>(sl.H, sl.h, Rnew[sl.te * sRn], xs.m, xs.P, mean, var);
        mean_out[sl.te] = mean;
        var_out[sl.te] = var;
        State<D> xf;  // filtered state before this step
        if (r == r0) {
            xf = carry;
        } else {
            int i = (int)(r - r0) - 1;
            int L0_ = L0;
            load_state<D>(xf, [=](int k) { return fs[fs_index(c, i, k, L0_, Dim<D>::NS)]; });
        }
        double mp[D], Pp[D * D];
        copy_n<D>(xf.m, mp);
        copy_n<D * D>(xf.P, Pp);
        predict<D>(sl.A, sl.a, sl.Q, mp, Pp);
        double G[D * D], g[D], L[D * D];
        ok = invert_dynamics<D>(xf.m, xf.P, mp, Pp, sl.A, G, g, L) && ok;
        predict<D>(G, g, L, xs.m, xs.P);
    }
    return ok ? 0 : 1;
}

// ------------------------------------------------------------------------------------------ affine passes
// Prior marginals (COV, !RAND): x' = A x + a, P' = A P A' + Q; emits N(H x + h, H P H' + R).
// rand (RAND, !COV): x' = A x + a + chol(Q + 1e-9 I).U' eps_t ; y = H x' + h + sqrt(R) eps_e.
template <int D> TGP_HD bool noise_factor(const double* Q, double* Lq) {  // lower factor, column-major
    double Qj[D * D], U[D * D];
    copy_n<D * D>(Q, Qj);
    TGP_UNROLL for (int i = 0; i < D; ++i) Qj[i + i * D] += 1e-9;  // lgc.jl:86
    bool ok = chol_upper<D>(Qj, U);
    TGP_UNROLL for (int j = 0; j < D; ++j) TGP_UNROLL for (int i = 0; i < D; ++i) Lq[i + j * D] = U[j + i * D];
    return ok;
}

template <int D, bool LTI, bool RAND, typename Store>
TGP_HD int chunk_reduce_affine(const ModelView& mv, int64_t c, int L0, const double* eps_t, Store st) {
    int64_t r0 = c * (int64_t)L0;
    int64_t r1 = r0 + L0 < mv.T ? r0 + L0 : mv.T;
    AElem<D> e;
    e.identity();
    StepLoader<D, LTI> sl;
    sl.init(mv);
    double Lq[D * D];
    bool ok = true;
    if (RAND && LTI) ok = noise_factor<D>(sl.Q, Lq);
    for (int64_t r = r0; r < r1; ++r) {
        sl.index(mv, r);
        sl.load_transition(mv);
        if (!sl.do_predict) continue;
        if (RAND) {
            if (!LTI) ok = noise_factor<D>(sl.Q, Lq) && ok;
            double cvec[D];
            const double* ep = eps_t + sl.tt * D;
            TGP_UNROLL for (int i = 0; i < D; ++i) {
                double acc = sl.a[i];
                TGP_UNROLL for (int k = 0; k <= i; ++k) acc = fma(Lq[i + k * D], ep[k], acc);
                cvec[i] = acc;
            }
            a_extend<D, false>(e, sl.A, cvec, sl.Q);
        } else {
            a_extend<D, true>(e, sl.A, sl.a, sl.Q);
        }
    }
    store_aelem<D>(e, st);
    return ok ? 0 : 1;
}

template <int D, bool LTI, bool RAND>
TGP_HD int chunk_apply_affine(const ModelView& mv, int64_t c, int L0, State<D>& x, const double* eps_t, const double* eps_e,
                              double* mean_out, double* var_out) {
    int64_t r0 = c * (int64_t)L0;
    int64_t r1 = r0 + L0 < mv.T ? r0 + L0 : mv.T;
    StepLoader<D, LTI> sl;
    sl.init(mv);
    double Lq[D * D];
    bool ok = true;
    if (RAND && LTI) ok = noise_factor<D>(sl.Q, Lq);
    for (int64_t r = r0; r < r1; ++r) {
        sl.index(mv, r);
        sl.load_transition(mv);
        sl.load_emission(mv);
        if (sl.do_predict) {
            if (RAND) {
                if (!LTI) ok = noise_factor<D>(sl.Q, Lq) && ok;
                const double* ep = eps_t + sl.tt * D;
                double xn[D];
                mat_vec<D>(sl.A, x.m, xn);
                TGP_UNROLL for (int i = 0; i < D; ++i) {
                    double nz = 0.0;
                    TGP_UNROLL for (int k = 0; k <= i; ++k) nz = fma(Lq[i + k * D], ep[k], nz);
                    x.m[i] = (xn[i] + sl.a[i]) + nz;
                }
            } else {
                predict<D>(sl.A, sl.a, sl.Q, x.m, x.P);
            }
        }
        if (RAND) {
            double yy = 0.0;
            TGP_UNROLL for (int i = 0; i < D; ++i) yy = fma(sl.H[i], x.m[i], yy);
            mean_out[sl.te] = (yy + sl.h) + sqrt(sl.R) * eps_e[sl.te];   // lgc.jl:241-243
        } else {
            double mean, var;
            emit_scalar<D>(sl.H, sl.h, sl.R, x.m, x.P, mean, var);
            mean_out[sl.te] = mean;
            var_out[sl.te] = var;
        }
    }
    return ok ? 0 : 1;
}

}  // namespace tgp
