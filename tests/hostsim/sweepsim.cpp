// TEST INFRASTRUCTURE ONLY (never part of the product library, never a fallback).
// The sweep engine (temporalgps.jl_amd/csrc/tgp_sweep.hpp) run on the host: the product's own plan (tgp_sweep_plan.hpp) and the very
// functions its kernel calls per lane (tgp_sweep_body.hpp: forward_run, backward_run and the per-step arithmetic under them), with the
// wave's 64 lanes visited one after the other and the two cross-lane shifts done by hand.  What is NOT exercised here is k_sweep's own
// orchestration (tgp_sweep.hip), which this file restates; the GPU tier covers it.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../temporalgps.jl_amd/csrc/tgp_sweep_plan.hpp"

using namespace tgp_sweep;

namespace {

template <int D> bool finite_state(const State<D>& x) {
    double s = 0.0;
    for (int k = 0; k < D; ++k) s += std::fabs(x.m[k]);
    for (int k = 0; k < SD<D>::DS; ++k) s += std::fabs(x.P[k]);
    return s < 1e300;
}

template <int D, bool SDE, int XS> void run_d(const Plan& p, const Streams& st, bool post, double* mean, double* var, double* out) {
    constexpr int B = Geo<D>::B, NS = SD<D>::NS;
    KArgs<D> ka;
    std::memcpy(&ka.mc, p.mc, sizeof ka.mc);
    ka.st = st;
    ka.T = p.T;
    ka.C = p.C;
    ka.W = p.W;
    ka.Wb = p.Wb;
    ka.nchunks = p.nchunks;
    ka.mean = mean;
    ka.var = var;
    const int C = p.C;
    const long long T = p.T;
    std::vector<double> ckpt((size_t)(C / B) * NS * 64), sF((size_t)B * NS * 64);
    double lml_total = 0.0, dfw = 0.0, dbw = 0.0;
    unsigned bits_total = 0;
    ModelR<D, SDE> mr;
    mr.init(ka.mc);
    State<D> gen, x0;
    set_state<D>(gen, ka.mc.gm, ka.mc.gP);
    set_state<D>(x0, ka.mc.x0m, ka.mc.x0P);
    for (long long wave = 0; wave < p.nwaves; ++wave) {
        long long t0[64], t1[64], t1r[64];
        bool runs[64], owned[64], ok[64];
        State<D> e1[64], x[64], b1[64], xs[64];
        LmlAcc acc[64];
        for (int lane = 0; lane < 64; ++lane) {
            const long long c = wave * kOwned + lane - 1;
            const bool active = c >= 0 && c < p.nchunks;
            t0[lane] = active ? c * C : 0;
            long long e = active ? t0[lane] + C : 0;
            t1[lane] = e < T ? e : (active ? T : 0);
            t1r[lane] = (t1[lane] + 7) & ~7ll;
            runs[lane] = active && lane >= 1;
            owned[lane] = runs[lane] && lane <= kOwned;
            ok[lane] = true;
        }
        // forwards, pass 0
        for (int lane = 0; lane < 64; ++lane) {
            const long long tw = t1r[lane] - ka.W;
            State<D> s = tw <= 0 ? x0 : gen;
            LmlAcc dummy;
            forward_run<D, SDE, XS, B>(ka, mr, tw, ka.W / B, tw > 0 ? tw : 0, t1r[lane], s, dummy, false, (double*)nullptr, lane, ok[lane]);
            e1[lane] = s;
        }
        // the shift, pass 1
        for (int lane = 0; lane < 64; ++lane) {
            x[lane] = lane > 0 ? e1[lane - 1] : e1[0];
            if (t0[lane] == 0) x[lane] = x0;
            acc[lane] = LmlAcc();
            const long long te = (t0[lane] + ka.Wb < t1[lane]) ? t0[lane] + ka.Wb : t1[lane];
            RevAcc<D> rev;
            rev.reset();
            b1[lane] = gen;
            const int nwin = post ? ka.Wb / B : 0;
            const long long hi1 = runs[lane] ? t1r[lane] : t0[lane];
            if (post) forward_run<D, SDE, XS, B, true>(ka, mr, t0[lane], nwin, t0[lane], hi1, x[lane], acc[lane], true, ckpt.data(), lane, ok[lane], &rev, t0[lane], te, &b1[lane]);
            forward_run<D, SDE, XS, B>(ka, mr, t0[lane] + (long long)nwin * B, C / B - nwin, t0[lane], hi1, x[lane], acc[lane], true,
                                       post ? ckpt.data() + (size_t)nwin * NS * 64 : (double*)nullptr, lane, ok[lane]);
        }
        double df[64] = {}, db[64] = {};
        bool fin[64];
        for (int lane = 0; lane < 64; ++lane) {
            fin[lane] = true;
            if (owned[lane]) {
                df[lane] = state_distance<D>(ka.mc, x[lane], e1[lane]);
                fin[lane] = finite_state<D>(x[lane]) && finite_state<D>(e1[lane]);
            }
        }
        if (post) {
            for (int lane = 0; lane < 64; ++lane) {
                xs[lane] = lane < 63 ? b1[lane + 1] : b1[63];
                backward_run<D, SDE, XS, B>(ka, mr, t0[lane], C / B, owned[lane] ? t1[lane] : t0[lane], t1[lane] == T, xs[lane], true, ckpt.data(), sF.data(),
                                            lane, ok[lane]);
                if (owned[lane]) {
                    db[lane] = state_distance<D>(ka.mc, xs[lane], b1[lane]);
                    fin[lane] = fin[lane] && finite_state<D>(xs[lane]) && finite_state<D>(b1[lane]);
                }
            }
        }
        double lml = 0.0;
        unsigned bits = 0;
        for (int lane = 0; lane < 64; ++lane) {
            const double l = owned[lane] ? acc[lane].total() : 0.0;
            lml += l;
            if (!(std::fabs(l) < 1e300)) fin[lane] = false;
            if (owned[lane] && !(df[lane] <= ka.mc.tol)) bits |= 1u;
            if (owned[lane] && post && !(db[lane] <= ka.mc.tol_b)) bits |= 2u;
            if (!ok[lane]) bits |= 4u;
            if (!fin[lane]) bits |= 8u;
            dfw = std::max(dfw, df[lane]);
            dbw = std::max(dbw, db[lane]);
        }
        lml_total += lml;
        bits_total |= bits;
    }
    out[0] = lml_total;
    out[1] = (double)bits_total;
    out[2] = dfw;
    out[3] = dbw;
}

template <int D, bool SDE> void run_x(const Plan& p, const Streams& st, bool post, double* mean, double* var, double* out) {
    switch ((st.R != nullptr ? 1 : 0) | (st.hh != nullptr ? 2 : 0)) {
        case 0: run_d<D, SDE, 0>(p, st, post, mean, var, out); break;
        case 1: run_d<D, SDE, 1>(p, st, post, mean, var, out); break;
        case 2: run_d<D, SDE, 2>(p, st, post, mean, var, out); break;
        default: run_d<D, SDE, 3>(p, st, post, mean, var, out); break;
    }
}
template <int D> void run_s(const Plan& p, const Streams& st, bool post, double* mean, double* var, double* out) {
    if (p.sde) run_x<D, true>(p, st, post, mean, var, out);
    else run_x<D, false>(p, st, post, mean, var, out);
}

}  // namespace

extern "C" int sweepsim_run(int d, int sde, int64_t T, const double* A, const double* a, const double* Q, const double* H, double hh, double R,
                            const double* x0m, const double* x0P, const double* coef, double tau_typ, const double* y, const uint8_t* mask,
                            const double* Rstep, const double* hstep, const double* tau, const double* Rnew, int rnew_per_step, int want_post, int fC,
                            int fW, int fWb, int w_hint, int wb_hint, int num_cu, double* mean, double* var, double* out) {
    ModelHost m;
    m.d = d;
    m.sde = sde != 0;
    m.A = A; m.a = a; m.Q = Q; m.H = H; m.hh = hh; m.R = R; m.x0m = x0m; m.x0P = x0P; m.coef = coef; m.tau_typ = tau_typ;
    Plan p;
    Forced f;
    f.C = fC; f.W = fW; f.Wb = fWb;
    std::string why;
    if (!make_plan(&p, f, m, T, w_hint, wb_hint, num_cu, &why)) return 1;
    Streams st;
    st.y = y; st.mask = mask; st.R = Rstep; st.hh = hstep; st.tau = tau; st.Rnew = Rnew; st.rnew_per_step = rnew_per_step;
    switch (d) {
        case 1: run_s<1>(p, st, want_post != 0, mean, var, out); break;
        case 2: run_s<2>(p, st, want_post != 0, mean, var, out); break;
        case 3: run_s<3>(p, st, want_post != 0, mean, var, out); break;
        default: run_s<4>(p, st, want_post != 0, mean, var, out); break;
    }
    out[4] = p.C;
    out[5] = p.W;
    out[6] = p.Wb;
    out[7] = (double)p.nwaves;
    return 0;
}
