// TEST INFRASTRUCTURE ONLY (never part of the product library, never a fallback).
// Runs the engine's per-lane chunk functions and scan monoids (temporalgps.jl_amd/csrc/tgp_math.hpp,
// tgp_chunk.hpp -- the very same headers the HIP kernels instantiate) sequentially on the host, with a
// small block size so that multi-level scans are exercised at tiny T. This lets the CPU-only test tier
// check the time-parallel algorithm (elements, combines, chunk hand-offs, scratch indexing) against the
// oracle without a GPU; the GPU tier then only has to establish that the HIP launch / shuffle / LDS
// mechanics reproduce it.
#include <cstdint>
#include <cstring>
#include <vector>
#include <cstdio>
#include <algorithm>
#include <cstdlib>

#include "../../temporalgps.jl_amd/csrc/tgp_chunk.hpp"

using namespace tgp;

namespace {

struct SoA {  // [ncomp][n]
    std::vector<double> v;
    int64_t n = 0;
    int nc = 0;
    void init(int nc_, int64_t n_) { nc = nc_; n = n_; v.assign((size_t)nc * n, 0.0); }
};

template <int D> struct FM {
    using E = FElem<D>;
    static constexpr int NC = Dim<D>::NF;
    static void load(E& e, const SoA& s, int64_t i) { load_felem<D>(e, [&](int k) { return s.v[(size_t)k * s.n + i]; }); }
    static void combine(const E& a, const E& b, E& o) { f_combine<D>(a, b, o); }
    static void apply(const E& e, const State<D>& in, State<D>& out) { f_apply<D>(e, in, out); }
};
template <int D, bool COV> struct AM {
    using E = AElem<D>;
    static constexpr int NC = Dim<D>::NA;
    static void load(E& e, const SoA& s, int64_t i) { load_aelem<D>(e, [&](int k) { return s.v[(size_t)k * s.n + i]; }); }
    static void combine(const E& a, const E& b, E& o) { a_combine<D, COV>(a, b, o); }
    static void apply(const E& e, const State<D>& in, State<D>& out) { a_apply<D, COV>(e, in, out); }
};

template <int D> void store_elem(const FElem<D>& e, SoA& s, int64_t i) {
    store_felem<D>(e, [&](int k, double v) { s.v[(size_t)k * s.n + i] = v; });
}
template <int D> void store_elem(const AElem<D>& e, SoA& s, int64_t i) {
    store_aelem<D>(e, [&](int k, double v) { s.v[(size_t)k * s.n + i] = v; });
}

// hierarchical scan, mirrors the GPU structure: reduce blocks upward, scan the top block, apply downward.
template <int D, class M> void hier_scan(const SoA& E0, const State<D>& x0, std::vector<State<D>>& S0, State<D>& fin, int BS) {
    std::vector<SoA> E;
    E.push_back(E0);
    while (E.back().n > BS) {
        const SoA& lo = E.back();
        SoA hi;
        hi.init(M::NC, (lo.n + BS - 1) / BS);
        for (int64_t b = 0; b < hi.n; ++b) {
            typename M::E acc, cur, tmp;
            acc.identity();
            for (int64_t i = b * BS; i < lo.n && i < (b + 1) * BS; ++i) {
                M::load(cur, lo, i);
                M::combine(acc, cur, tmp);
                acc = tmp;
            }
            store_elem<D>(acc, hi, b);
        }
        E.push_back(hi);
    }
    std::vector<std::vector<State<D>>> S(E.size());
    for (int l = (int)E.size() - 1; l >= 0; --l) {
        S[l].resize(E[l].n);
        int64_t nb = (l == (int)E.size() - 1) ? 1 : E[l + 1].n;
        int64_t bs = (l == (int)E.size() - 1) ? E[l].n : BS;
        for (int64_t b = 0; b < nb; ++b) {
            State<D> carry = (l == (int)E.size() - 1) ? x0 : S[l + 1][b];
            typename M::E acc, cur, tmp;
            acc.identity();
            for (int64_t i = b * bs; i < E[l].n && i < (b + 1) * bs; ++i) {
                M::apply(acc, carry, S[l][i]);
                M::load(cur, E[l], i);
                M::combine(acc, cur, tmp);
                acc = tmp;
            }
            if (l == (int)E.size() - 1) M::apply(acc, carry, fin);
        }
    }
    S0 = S[0];
}

template <int D> State<D> make_state(const double* m, const double* P) {
    State<D> s;
    for (int i = 0; i < D; ++i) s.m[i] = m[i];
    for (int i = 0; i < D * D; ++i) s.P[i] = P[i];
    return s;
}

struct Args {
    ModelView mv;
    const double *x0m, *x0P;
    int L0, BS;
    // outputs / extra inputs
    double* lml;
    double *m_out, *P_out;
    double *G_out, *g_out, *L_out, *xfm, *xfP;
    const double* Rnew;
    int64_t sRn;
    double *mean_out, *var_out;
    const double *eps_t, *eps_e;
    int what;  // 0 logpdf, 1 filter, 2 posterior(+marginals if mean_out), 3 prior marginals, 4 rand, 5 segment reduce
    double *elem_out, *rev_out;   // total filter element of the segment / total smoother element
    const double *xs_m, *xs_P;    // smoothed state at the segment end (null: the final filtered state)
};

template <int D, class M> void total_elem(const SoA& E0, double* out) {
    typename M::E acc, cur, tmp;
    acc.identity();
    for (int64_t i = 0; i < E0.n; ++i) {
        M::load(cur, E0, i);
        M::combine(acc, cur, tmp);
        acc = tmp;
    }
    SoA one;
    one.init(M::NC, 1);
    store_elem<D>(acc, one, 0);
    for (int k = 0; k < M::NC; ++k) out[k] = one.v[k];
}

template <int D, bool LTI> int run(const Args& a) {
    ModelView mv = a.mv;
    int64_t n0 = (mv.T + a.L0 - 1) / a.L0;
    std::vector<double> tile_t, tile_e;
    // HOSTSIM_STEADY=1: the stationary-covariance steps of passes 2 / 3 (tgp_chunk_body.inc), as tgp_api.hip enables them
    std::vector<double> steady_rec;
    constexpr bool HS = LTI && D <= kSteadyMaxD;     // the build of passes 2 / 3 with the stationary-covariance steps exists
    const char* steady_env = getenv("HOSTSIM_STEADY");
    if (steady_env != nullptr && LTI && D <= kSteadyMaxD && mv.p == 1 && mv.sR == 0 && mv.missing == nullptr) {
        steady_rec.assign((size_t)(1 + D * (D + 1)) * (size_t)n0, 0.0);
        mv.steady = steady_rec.data();
    }
    if (!LTI) {   // general layout: time-tiled copies of the per-step arrays, as tgp_api.hip builds on the device
        mv.tile_mask = tile_mask_of(mv);
        mv.nc_t = tile_offset_t(mv.tile_mask, 0u, D);
        mv.nc_e = tile_offset_e(mv.tile_mask, 0u, D);
        const int Lt = a.L0 / mv.p;
        const size_t nblk = (size_t)((n0 + 63) / 64) * 64;
        tile_t.assign(nblk * Lt * mv.nc_t + 1, 0.0);
        tile_e.assign(nblk * a.L0 * mv.nc_e + 1, 0.0);
        for (int64_t c = 0; c < n0; ++c) {
            for (int tl = 0; tl < Lt; ++tl) tile_transition(a.mv, D, mv.tile_mask, mv.nc_t, Lt, c, tl, tile_t.data());
            for (int i = 0; i < a.L0; ++i) tile_emission(a.mv, D, mv.tile_mask, mv.nc_e, a.L0, c, i, tile_e.data());
        }
        mv.tile_t = tile_t.data();
        mv.tile_e = tile_e.data();
    }
    State<D> x0 = make_state<D>(a.x0m, a.x0P);
    int bad = 0;
    if (a.what <= 2 || a.what == 5) {
        SoA E0;
        E0.init(Dim<D>::NF, n0);
        // (with -fopenmp -- the all-core CPU baseline build, oracle/omp_scan.py -- the three O(T) chunk loops run one chunk per
        //  thread at a time; the block scans over the n0 chunk elements stay sequential)
#pragma omp parallel for schedule(static)
        for (int64_t c = 0; c < n0; ++c)
        {
            DirectIO io{mv.y, mv.R, nullptr, nullptr};
            chunk_reduce_filter<D, LTI>(mv, c, a.L0, io, [&](int k, double v) { E0.v[(size_t)k * n0 + c] = v; });
        }
        if (a.elem_out) total_elem<D, FM<D>>(E0, a.elem_out);
        if (a.what == 5) return 0;
        std::vector<State<D>> S0;
        State<D> fin;
        hier_scan<D, FM<D>>(E0, x0, S0, fin, a.BS);
        double lml = 0.0, nmiss = 0.0;
        FilterOut fo{a.m_out, a.P_out, nullptr, a.G_out, a.g_out, a.L_out};
        std::vector<double> fs;
        SoA R0;
        if (a.what == 2) {
            fs.assign((size_t)((n0 + 63) / 64) * 64 * a.L0 * Dim<D>::NS, 0.0);
            fo.fs = fs.data();
            R0.init(Dim<D>::NA, n0);
        }
        std::vector<double> lml_c((size_t)n0, 0.0), nmiss_c((size_t)n0, 0.0);
        std::vector<int> bad_c((size_t)n0, 0);
        std::vector<double> xfin_buf(Dim<D>::NS, 0.0);   // Reverse prior, MODE 3: x0 of the posterior (written by the last chunk)
#pragma omp parallel for schedule(static)
        for (int64_t c = 0; c < n0; ++c) {
            State<D> x = S0[c];
            ChunkStats cs;
            auto nost = [](int, double) {};
            DirectIO io{mv.y, mv.R, nullptr, nullptr};
            if (a.what == 0) cs = mv.steady ? chunk_apply_filter<D, LTI, 0, HS>(mv, c, a.L0, x, fo, io, nost) : chunk_apply_filter<D, LTI, 0>(mv, c, a.L0, x, fo, io, nost);
            else if (a.what == 1) cs = mv.steady ? chunk_apply_filter<D, LTI, 1, HS>(mv, c, a.L0, x, fo, io, nost) : chunk_apply_filter<D, LTI, 1>(mv, c, a.L0, x, fo, io, nost);
            else {
                if (a.G_out) {   // materialised posterior model (MODE 3) ...
                    State<D> x3 = x;
                    FilterOut f3{nullptr, nullptr, nullptr, a.G_out, a.g_out, a.L_out, xfin_buf.data()};
                    ChunkStats c3 = chunk_apply_filter<D, LTI, 3>(mv, c, a.L0, x3, f3, io, nost);
                    bad_c[c] |= c3.bad;
                }
                FilterOut f2{nullptr, nullptr, fo.fs, nullptr, nullptr, nullptr};   // ... and the smoother forward pass (MODE 2)
                auto rst = [&](int k, double v) { R0.v[(size_t)k * n0 + (n0 - 1 - c)] = v; };
                cs = mv.steady ? chunk_apply_filter<D, LTI, 2, HS>(mv, c, a.L0, x, f2, io, rst) : chunk_apply_filter<D, LTI, 2>(mv, c, a.L0, x, f2, io, rst);
            }
            lml_c[c] = cs.lml;
            nmiss_c[c] = cs.nmiss;
            bad_c[c] |= cs.bad;
        }
        for (int64_t c = 0; c < n0; ++c) {      // fixed summation order
            lml += lml_c[c];
            nmiss += nmiss_c[c];
            bad |= bad_c[c];
        }
        if (a.lml) *a.lml = lml + nmiss * 0.5 * (kLog2Pi + log(kLargeVar));
        if (a.xfm) {
            if (mv.ordering != 0 && a.G_out) load_state<D>(fin, [&](int k) { return xfin_buf[k]; });   // posterior x0 of a Reverse prior
            for (int i = 0; i < D; ++i) a.xfm[i] = fin.m[i];
            for (int i = 0; i < D * D; ++i) a.xfP[i] = fin.P[i];
        }
        if (a.what == 2 && a.rev_out) total_elem<D, AM<D, true>>(R0, a.rev_out);
        if (a.what == 2 && a.mean_out) {
            std::vector<State<D>> S0r;
            State<D> fin_r;
            State<D> seed = a.xs_m ? make_state<D>(a.xs_m, a.xs_P) : fin;
            hier_scan<D, AM<D, true>>(R0, seed, S0r, fin_r, a.BS);
            std::vector<int> bad_s((size_t)n0, 0);
#pragma omp parallel for schedule(static)
            for (int64_t c = 0; c < n0; ++c) {
                State<D> xs = S0r[n0 - 1 - c];
                DirectIO io{nullptr, a.Rnew, a.mean_out, a.var_out};
                bad_s[c] = mv.steady ? chunk_smooth<D, LTI, HS>(mv, c, a.L0, xs, S0[c], fs.data(), a.sRn, io)
                                     : chunk_smooth<D, LTI>(mv, c, a.L0, xs, S0[c], fs.data(), a.sRn, io);
            }
            for (int64_t c = 0; c < n0; ++c) bad |= bad_s[c];
            if (mv.steady != nullptr && steady_env[0] == '2') {   // how many steps of the series pass 2 ran in the mean-only form
                int64_t fast = 0;
                for (int64_t c = 0; c < n0; ++c) {
                    const int64_t len = std::min<int64_t>(a.L0, mv.T - c * (int64_t)a.L0), k = (int64_t)mv.steady[c];
                    if (k < len) fast += len - k;
                }
                fprintf(stderr, "hostsim steady: %lld of %lld steps mean-only\n", (long long)fast, (long long)mv.T);
            }
        }
    } else {
        SoA E0;
        E0.init(Dim<D>::NA, n0);
        for (int64_t c = 0; c < n0; ++c) {
            auto st = [&](int k, double v) { E0.v[(size_t)k * n0 + c] = v; };
            if (a.what == 3) bad |= chunk_reduce_affine<D, LTI, false>(mv, c, a.L0, nullptr, st);
            else bad |= chunk_reduce_affine<D, LTI, true>(mv, c, a.L0, a.eps_t, st);
        }
        std::vector<State<D>> S0;
        State<D> fin;
        if (a.what == 3) hier_scan<D, AM<D, true>>(E0, x0, S0, fin, a.BS);
        else hier_scan<D, AM<D, false>>(E0, x0, S0, fin, a.BS);
        for (int64_t c = 0; c < n0; ++c) {
            State<D> x = S0[c];
            DirectIO io{a.eps_e, mv.R, a.mean_out, a.what == 3 ? a.var_out : nullptr};
            if (a.what == 3) bad |= chunk_apply_affine<D, LTI, false>(mv, c, a.L0, x, nullptr, io);
            else bad |= chunk_apply_affine<D, LTI, true>(mv, c, a.L0, x, a.eps_t, io);
        }
    }
    return bad ? 2 : 0;
}

// ---- forward-mode gradient of logpdf w.r.t. ONE parameter (LTI models): the tgp::ad instantiation of the same
// passes, with a hierarchical scan over Dual elements that mirrors the device structure.
struct GradArgs {
    const double *dx0m, *dx0P;
    double* dlml;
};

template <int D> int run_grad(const Args& a, const GradArgs& ga) {
    namespace A = tgp::ad;
    const ModelView& mv = a.mv;
    const int64_t n0 = (mv.T + a.L0 - 1) / a.L0;
    std::vector<A::FElem<D>> E(n0);
    for (int64_t c = 0; c < n0; ++c) {
        DirectIO io{mv.y, mv.R, nullptr, nullptr};
        double buf[2 * Dim<D>::NF];
        A::chunk_reduce_filter<D, true>(mv, c, a.L0, io, [&](int k, Dual v) { buf[2 * k] = v.v; buf[2 * k + 1] = v.d; });
        A::load_felem<D>(E[c], [&](int k) { return Dual(buf[2 * k], buf[2 * k + 1]); });
    }
    A::State<D> x0;
    for (int i = 0; i < D; ++i) x0.m[i] = Dual(a.x0m[i], ga.dx0m ? ga.dx0m[i] : 0.0);
    for (int i = 0; i < D * D; ++i) x0.P[i] = Dual(a.x0P[i], ga.dx0P ? ga.dx0P[i] : 0.0);
    // two-level scan (blocks of BS): block totals, block carries, in-block carries
    std::vector<A::State<D>> S(n0);
    const int64_t nb = (n0 + a.BS - 1) / a.BS;
    std::vector<A::FElem<D>> B(nb);
    for (int64_t b = 0; b < nb; ++b) {
        A::FElem<D> acc, tmp;
        acc.identity();
        for (int64_t i = b * a.BS; i < n0 && i < (b + 1) * a.BS; ++i) { A::f_combine<D>(acc, E[i], tmp); acc = tmp; }
        B[b] = acc;
    }
    A::State<D> carry = x0, nxt;
    for (int64_t b = 0; b < nb; ++b) {
        A::FElem<D> acc, tmp;
        acc.identity();
        for (int64_t i = b * a.BS; i < n0 && i < (b + 1) * a.BS; ++i) {
            A::f_apply<D>(acc, carry, S[i]);
            A::f_combine<D>(acc, E[i], tmp);
            acc = tmp;
        }
        A::f_apply<D>(B[b], carry, nxt);
        carry = nxt;
    }
    Dual lml(0.0, 0.0);
    double nmiss = 0.0;
    int bad = 0;
    FilterOut fo{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    for (int64_t c = 0; c < n0; ++c) {
        DirectIO io{mv.y, mv.R, nullptr, nullptr};
        A::State<D> x = S[c];
        A::ChunkStats cs = A::chunk_apply_filter<D, true, 0>(mv, c, a.L0, x, fo, io, [](int, Dual) {});
        lml += cs.lml;
        nmiss += cs.nmiss;
        bad |= cs.bad;
    }
    if (a.lml) *a.lml = lml.v + nmiss * 0.5 * (kLog2Pi + log(kLargeVar));
    *ga.dlml = lml.d;
    return bad ? 2 : 0;
}

template <int D> int run_d(const Args& a, bool lti) { return lti ? run<D, true>(a) : run<D, false>(a); }

}  // namespace

extern "C" int hostsim_grad(int d, int L0, int BS, int64_t T, const double* A, const double* av, const double* Q, const double* H,
                            const double* h, const double* R, const double* y, const uint8_t* missing, const double* x0m,
                            const double* x0P, const double* dA, const double* da, const double* dQ, const double* dH, const double* dh,
                            const double* dR, const double* dx0m, const double* dx0P, double* lml, double* dlml) {
    Args a{};
    a.mv = ModelView{T, 0, 1, A, av, Q, H, h, R, 0, 0, 0, 0, 0, 0, y, missing, nullptr, nullptr, 0, 0, 0u, 0, T, dA, da, dQ, dH, dh, dR};
    a.x0m = x0m; a.x0P = x0P; a.L0 = L0; a.BS = BS; a.lml = lml;
    GradArgs ga{dx0m, dx0P, dlml};
    switch (d) {
        case 1: return run_grad<1>(a, ga);
        case 2: return run_grad<2>(a, ga);
        case 3: return run_grad<3>(a, ga);
        case 4: return run_grad<4>(a, ga);
        case 5: return run_grad<5>(a, ga);
        case 6: return run_grad<6>(a, ga);
        default: return 4;
    }
}

extern "C" int hostsim_run(int d, int p, int small_out, int lti, int what, int L0, int BS, int64_t T, int ordering, const double* A, int64_t sA,
                           const double* av, int64_t sa, const double* Q, int64_t sQ, const double* H, int64_t sH,
                           const double* h, int64_t sh, const double* R, int64_t sR, const double* y, const uint8_t* missing,
                           const double* x0m, const double* x0P, double* lml, double* m_out, double* P_out, double* G_out,
                           double* g_out, double* L_out, double* xfm, double* xfP, const double* Rnew, int64_t sRn,
                           double* mean_out, double* var_out, const double* eps_t, const double* eps_e, double* elem_out,
                           double* rev_out, const double* xs_m, const double* xs_P) {
    Args a;
    a.mv = ModelView{T * p, ordering, p, A, av, Q, H, h, R, sA, sa, sQ, sH, sh, sR, y, missing, nullptr, nullptr, 0, 0, 0u, small_out, T,
                     nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    a.x0m = x0m; a.x0P = x0P; a.L0 = L0; a.BS = BS; a.lml = lml; a.m_out = m_out; a.P_out = P_out;
    a.G_out = G_out; a.g_out = g_out; a.L_out = L_out; a.xfm = xfm; a.xfP = xfP; a.Rnew = Rnew; a.sRn = sRn;
    a.mean_out = mean_out; a.var_out = var_out; a.eps_t = eps_t; a.eps_e = eps_e; a.what = what;
    a.elem_out = elem_out; a.rev_out = rev_out; a.xs_m = xs_m; a.xs_P = xs_P;
    switch (d) {
        case 1: return run_d<1>(a, lti);
        case 2: return run_d<2>(a, lti);
        case 3: return run_d<3>(a, lti);
        case 4: return run_d<4>(a, lti);
        case 5: return run_d<5>(a, lti);
        case 6: return run_d<6>(a, lti);
        case 7: return run_d<7>(a, lti);
        case 8: return run_d<8>(a, lti);
        case 10: return run_d<10>(a, lti);
        case 14: return run_d<14>(a, lti);
        case 16: return run_d<16>(a, lti);
        default: return 4;
    }
}
