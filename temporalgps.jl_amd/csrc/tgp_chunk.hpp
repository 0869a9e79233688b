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

TGP_HD int64_t step_index(const ModelView& mv, int64_t r) { return mv.ordering == 0 ? r : mv.T - 1 - r; }

// Per-step SCALAR streams (y, per-step R, R_new in; mean/var out) are accessed through an IO object in
// groups of G consecutive steps. A lane's chunk is contiguous in time, so lane-wise access is strided by
// L0*8 bytes across the wave; the device IO (WaveIO, tgp_kernels.hpp) instead lets 8 lanes fetch / write
// one chunk's 64 contiguous bytes and transposes through wave-private LDS. DirectIO is the plain form
// (host emulation, and the reference semantics the staged form must reproduce).
struct DirectIO {
    static constexpr int G = 8;
    const double* a0;  // y
    const double* a1;  // per-step R (or R_new); only read when its stride is non-zero
    double* o0;
    double* o1;
    TGP_HD void begin(const ModelView&, int64_t, int, int) {}
    TGP_HD double in0(int64_t te, int) const { return a0[te]; }
    TGP_HD double in1(int64_t te, int) const { return a1[te]; }
    TGP_HD void out(int64_t te, int, double x0, double x1) {
        o0[te] = x0;
        if (o1) o1[te] = x1;
    }
    TGP_HD void flush(const ModelView&, int64_t, int, int) {}
};

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
    // emission H, h (R separately: it may come through the staged IO)
    TGP_HD void load_emission(const ModelView& mv) {
        if (!LTI) {
            const double* pH = mv.H + te * mv.sH;
            TGP_UNROLL for (int i = 0; i < D; ++i) H[i] = pH[i];
            h = mv.h[te * mv.sh];
        }
    }
    TGP_HD void load_R_direct(const ModelView& mv) { R = mv.R[te * mv.sR]; }
    // y via io.in0, per-step R via io.in1 (io.a1 == mv.R); shared R read once per step from mv.R[0]
    template <class IO> TGP_HD void load_obs(const ModelView& mv, const IO& io, int i) {
        R = (mv.sR == 0) ? mv.R[0] : io.in1(te, i);
        y = io.in0(te, i);
        is_missing = (mv.missing != nullptr) && (mv.missing[te] != 0);
        if (is_missing) { y = 0.0; R = kLargeVar; }
    }
    template <class IO> TGP_HD void load(const ModelView& mv, int64_t r, const IO& io, int i) {
        index(mv, r);
        load_transition(mv);
        load_emission(mv);
        load_obs(mv, io, i);
    }
};

// ------------------------------------------------------------------------------------------ pass 1
// Chunk bounds; lanes past the last chunk (c >= n0) get an empty range but still take part in the
// wave-cooperative IO.
TGP_HD void chunk_range(const ModelView& mv, int64_t c, int L0, int64_t& r0, int64_t& r1) {
    r0 = c * (int64_t)L0;
    if (r0 > mv.T) r0 = mv.T;
    r1 = r0 + L0 < mv.T ? r0 + L0 : mv.T;
}

template <int D, bool LTI, class IO, typename Store>
TGP_HD void chunk_reduce_filter(const ModelView& mv, int64_t c, int L0, IO& io, Store st) {
    int64_t r0, r1;
    chunk_range(mv, c, L0, r0, r1);
    FElem<D> e;
    e.identity();
    StepLoader<D, LTI> sl;
    sl.init(mv);
    for (int g = 0; g < L0; g += IO::G) {
        io.begin(mv, c, g, L0);
        const int64_t rg = r0 + g;
        const int gend = (int)((r1 - rg) < IO::G ? (r1 - rg) : IO::G);
        for (int i = 0; i < gend; ++i) {
            sl.load(mv, rg + i, io, i);
            f_extend<D>(e, sl.do_predict, sl.A, sl.a, sl.Q, sl.H, sl.h, sl.R, sl.y);
        }
    }
    if (r1 > r0) store_felem<D>(e, st);
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
template <int D, bool LTI, int MODE, class IO, typename RStore>
TGP_HD ChunkStats chunk_apply_filter(const ModelView& mv, int64_t c, int L0, State<D>& x, const FilterOut& out, IO& io, RStore rst) {
    int64_t r0, r1;
    chunk_range(mv, c, L0, r0, r1);
    ChunkStats cs{0.0, 0.0, 0};
    StepLoader<D, LTI> sl;
    sl.init(mv);
    AElem<D> rev;
    if (MODE == 2) rev.identity();
    bool ok = true;
    for (int g = 0; g < L0; g += IO::G) {
      io.begin(mv, c, g, L0);
      const int64_t rg = r0 + g;
      const int gend = (int)((r1 - rg) < IO::G ? (r1 - rg) : IO::G);
      for (int gi = 0; gi < gend; ++gi) {
        const int64_t r = rg + gi;
        sl.load(mv, r, io, gi);
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
    }
    if (MODE == 2 && r1 > r0) store_aelem<D>(rev, rst);
    cs.bad = ok ? 0 : 1;
    return cs;
}

// ------------------------------------------------------------------------------------------ pass 3
// RTS smoother on chunk c. `xs` = smoothed state at the chunk's LAST step; `carry` = filtered state
// just before the chunk's first step. Emits N(H x + h, H P H' + Rnew) for every step (lgssm.jl:111-115
// on the posterior model with replace_observation_noise_cov, missings.jl:35-41).
template <int D, bool LTI, class IO>
TGP_HD int chunk_smooth(const ModelView& mv, int64_t c, int L0, State<D>& xs, const State<D>& carry, const double* fs,
                        int64_t sRn, IO& io) {
    int64_t r0, r1;
    chunk_range(mv, c, L0, r0, r1);
    StepLoader<D, LTI> sl;
    sl.init(mv);
    bool ok = true;
    const double Rn_shared = (sRn == 0) ? io.a1[0] : 0.0;
    for (int g = ((L0 - 1) / IO::G) * IO::G; g >= 0; g -= IO::G) {
        io.begin(mv, c, g, L0);   // stages R_new (io.a1) when it is per-step
        const int64_t rg = r0 + g;
        const int gend = (int)((r1 - rg) < IO::G ? (r1 - rg) : IO::G);
        for (int gi = gend - 1; gi >= 0; --gi) {
            const int64_t r = rg + gi;
            sl.index(mv, r);
            sl.load_transition(mv);
            sl.load_emission(mv);
            double mean, var;
            emit_scalar<D>(sl.H, sl.h, (sRn == 0) ? Rn_shared : io.in1(sl.te, gi), xs.m, xs.P, mean, var);
            io.out(sl.te, gi, mean, var);
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
            double G[D * D], g_[D], L[D * D];
            ok = invert_dynamics<D>(xf.m, xf.P, mp, Pp, sl.A, G, g_, L) && ok;
            predict<D>(G, g_, L, xs.m, xs.P);
        }
        io.flush(mv, c, g, L0);
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
    int64_t r0, r1;
    chunk_range(mv, c, L0, r0, r1);
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

template <int D, bool LTI, bool RAND, class IO>
TGP_HD int chunk_apply_affine(const ModelView& mv, int64_t c, int L0, State<D>& x, const double* eps_t, IO& io) {
    int64_t r0, r1;
    chunk_range(mv, c, L0, r0, r1);
    StepLoader<D, LTI> sl;
    sl.init(mv);
    double Lq[D * D];
    bool ok = true;
    if (RAND && LTI) ok = noise_factor<D>(sl.Q, Lq);
    for (int g = 0; g < L0; g += IO::G) {
      io.begin(mv, c, g, L0);   // RAND: stages eps_e (io.a0); per-step R (io.a1)
      const int64_t rg = r0 + g;
      const int gend = (int)((r1 - rg) < IO::G ? (r1 - rg) : IO::G);
      for (int gi = 0; gi < gend; ++gi) {
        const int64_t r = rg + gi;
        sl.index(mv, r);
        sl.load_transition(mv);
        sl.load_emission(mv);
        const double R = (mv.sR == 0) ? mv.R[0] : io.in1(sl.te, gi);
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
            io.out(sl.te, gi, (yy + sl.h) + sqrt(R) * io.in0(sl.te, gi), 0.0);   // lgc.jl:241-243
        } else {
            double mean, var;
            emit_scalar<D>(sl.H, sl.h, R, x.m, x.P, mean, var);
            io.out(sl.te, gi, mean, var);
        }
      }
      io.flush(mv, c, g, L0);
    }
    return ok ? 0 : 1;
}

}  // namespace tgp
