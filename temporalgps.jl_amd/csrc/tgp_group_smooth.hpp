// Posterior path in the group layout (G lanes per chunk, lane j = column j; LTI family, scalar observations, Forward):
//   k_group_apply_posterior   pass 2, MODE 2: filter from the carry-in state, filtered states to scratch, and the chunk's
//                             smoother element composed step by step from the reference's (jittered) invert_dynamics
//   k_group_smooth            pass 3: RTS recursion inside the chunk from the smoothed state at its end, emitting
//                             N(H x + h, H P H' + R_new) per step
// Same recursions as chunk_apply_filter<MODE 2> / chunk_smooth (tgp_chunk_body.inc) and invert_dynamics_impl /
// a_extend_right_impl (tgp_math_body.inc). Column-distributed pieces on top of tgp_group.hpp / tgp_group_scan.hpp:
//   Cholesky   upper factor U of Pp + 1e-10 I by rows: at step i lane i finishes U[i][i] and broadcasts its column above the
//              diagonal, the diagonal and its reciprocal; every lane j > i then has U[i][j]
//   solves     U is published once; the two triangular solves for column j of (U'U)^-1 (A Pf) are lane-local
//   G = Gt'    one transpose through the tile; L = Pf - (U Gt)'(U Gt) with one more publish
#pragma once

namespace TGP_NS {

// scratch of filtered states in the group layout: [chunk][step][packed state]
template <int D> __device__ __forceinline__ int64_t gfs_index(int64_t c, int i, int L0) { return (c * (int64_t)L0 + i) * Dim<D>::NS; }
template <int D> __device__ __forceinline__ void gfs_store(double* __restrict__ fs, int64_t base, int j, double mj, const double* Pc) {
    fs[base + j] = mj;
    TGP_GUNROLL for (int i = 0; i < D; ++i)
        if (i <= j) fs[base + D + j * (j + 1) / 2 + i] = Pc[i];
}
template <int D> __device__ __forceinline__ void gfs_load(const double* __restrict__ fs, int64_t base, int j, bool act, double& mj, double* Pc) {
    mj = 0.0;
    TGP_GUNROLL for (int i = 0; i < D; ++i) Pc[i] = 0.0;
    if (!act) return;
    mj = fs[base + j];
    TGP_GUNROLL for (int i = 0; i < D; ++i) {
        const int lo = i < j ? i : j, hi = i < j ? j : i;
        Pc[i] = fs[base + D + hi * (hi + 1) / 2 + lo];
    }
}

// Alternative emission block for pass 3: marginals of N(Hn x_t + hn, Hn P_t Hn' + Rn) under the SMOOTHED state, for pn
// functionals per time step that are not the model's own observations (predictions at new outputs / locations).
struct GroupAltEmit {
    const double* H;      // [pn][d] or NULL (use the model's emissions)
    const double* h;      // [pn]
    const double* R;      // [T|1][pn]
    int64_t sR;           // 0: shared
    int pn;
};

template <int D> struct GroupRts {
    static constexpr int G = GroupGeom<D>::G, V0 = GroupGeom<D>::V0, V1 = GroupGeom<D>::V1;
    GroupOps<D> op;
    const double* sA;     // A in LDS, row-major [G i + k]

    __device__ __forceinline__ void mul_A(const double* x, double* y) const {
        TGP_GUNROLL for (int i = 0; i < D; ++i) {
            double acc = 0.0;
            TGP_GUNROLL for (int k = 0; k < D; ++k) acc = fma(sA[G * i + k], x[k], acc);
            y[i] = acc;
        }
    }
    // m <- A m + a ; P <- A P A' + Q     (GroupLane::predict)
    __device__ __forceinline__ void predict(double& vj, double aj, double* Sc, const double* Qc) const {
        double W[D], row[D], v[D];
        mul_A(Sc, W);
        wave_sync();
        TGP_GUNROLL for (int i = 0; i < D; ++i) op.tile[i + G * op.j] = W[i];
        op.tile[V0 + op.j] = vj;
        wave_sync();
        TGP_GUNROLL for (int k = 0; k < D; ++k) {
            row[k] = op.act ? op.tile[op.j + G * k] : 0.0;
            v[k] = op.tile[V0 + k];
        }
        mul_A(row, Sc);
        TGP_GUNROLL for (int i = 0; i < D; ++i) Sc[i] += Qc[i];
        double acc = 0.0;
        TGP_GUNROLL for (int k = 0; k < D; ++k) acc = fma(sA[G * op.j + k], v[k], acc);
        vj = acc + aj;
    }

    // invert_dynamics (lgssm.jl:231-238): filtered (mf, Pf), predicted (mp, Pp) -> x_{k-1} | x_k ~ N(G x_k + g, L).
    // Outputs by columns: Gc = column j of G, Xc = column j of Gt = G' (i.e. ROW j of G), gj, Lc.
    __device__ __forceinline__ bool invert_dynamics(double mfj, const double* Pfc, double mpj, const double* Ppc, double* Gc, double* Xc,
                                                    double& gj, double* Lc) const {
        const int j = op.j;
        double Uc[D], rinv[D];
        bool ok = true;
        // ---- Cholesky of Pp + 1e-10 I (upper factor, U'U), one row per step
        TGP_GUNROLL for (int i = 0; i < D; ++i) Uc[i] = 0.0;
        TGP_GUNROLL for (int i = 0; i < D; ++i) {
            // lane i: its column above the diagonal is complete; finish the diagonal and hand both to everybody
            double acc = Ppc[i] + ((i == j) ? 1e-10 : 0.0);          // S[i][j] (row i of this lane's column)
            wave_sync();
            if (j == i) {
                double dg = acc;
                TGP_GUNROLL for (int k = 0; k < i; ++k) dg = fma(-Uc[k], Uc[k], dg);
                const double di = sqrt(dg);
                TGP_GUNROLL for (int k = 0; k < i; ++k) op.tile[V0 + k] = Uc[k];
                op.tile[V1 + 0] = dg;
                op.tile[V1 + 1] = di;
                op.tile[V1 + 2] = 1.0 / di;
            }
            wave_sync();
            const double dg = op.tile[V1 + 0], di = op.tile[V1 + 1];
            rinv[i] = op.tile[V1 + 2];
            ok = ok && (dg > 0.0);
            TGP_GUNROLL for (int k = 0; k < i; ++k) acc = fma(-op.tile[V0 + k], Uc[k], acc);   // - sum_k U[k][i] U[k][j]
            Uc[i] = (j == i) ? di : ((j > i && op.act) ? acc * rinv[i] : 0.0);
        }
        // ---- Gt = U \\ (U' \\ (A Pf)), column j
        double Bc[D];
        mul_A(Pfc, Bc);
        op.publish(Uc);                                              // tile[i + G k] = U[i][k]
        TGP_GUNROLL for (int i = 0; i < D; ++i) {                    // U' z = b
            double acc = Bc[i];
            TGP_GUNROLL for (int k = 0; k < i; ++k) acc = fma(-op.tile[k + G * i], Xc[k], acc);
            Xc[i] = acc * rinv[i];
        }
        TGP_GUNROLL for (int i = D - 1; i >= 0; --i) {               // U w = z
            double acc = Xc[i];
            TGP_GUNROLL for (int k = i + 1; k < D; ++k) acc = fma(-op.tile[i + G * k], Xc[k], acc);
            Xc[i] = acc * rinv[i];
        }
        if (!op.act) { TGP_GUNROLL for (int i = 0; i < D; ++i) Xc[i] = 0.0; }
        // ---- UG = U Gt (U still published), then L = Pf - UG' UG
        double UG[D];
        TGP_GUNROLL for (int i = 0; i < D; ++i) {
            double acc = 0.0;
            TGP_GUNROLL for (int k = i; k < D; ++k) acc = fma(op.tile[i + G * k], Xc[k], acc);
            UG[i] = acc;
        }
        // g = mf - G mp   with (G mp)_j = sum_k G[j][k] mp_k = sum_k Gt[k][j] mp_k
        double mp[D];
        op.gather(mpj, mp);
        double acc = 0.0;
        TGP_GUNROLL for (int k = 0; k < D; ++k) acc = fma(Xc[k], mp[k], acc);
        gj = mfj - acc;
        op.publish(UG);
        TGP_GUNROLL for (int i = 0; i < D; ++i) {
            double a2 = 0.0;
            TGP_GUNROLL for (int k = 0; k < D; ++k) a2 = fma(op.tile[k + G * i], UG[k], a2);
            Lc[i] = op.act ? Pfc[i] - a2 : 0.0;
        }
        // ---- G = Gt': column j of G is row j of Gt
        op.publish(Xc);
        op.row(Gc);
        return ok;
    }

    // e <- e o (G, g, L)     (a_extend_right_impl): e.g += E g ; e.L += E L E' ; e.E = E G
    __device__ __forceinline__ void extend_right(GAElem<D>& e, const double* Gc, double gj, const double* Lc) const {
        double t[D], v[D], T1[D], EG[D];
        op.gather(gj, v);
        op.publish(e.E);
        op.row(t);                                                   // row j of E
        double acc = 0.0;
        TGP_GUNROLL for (int k = 0; k < D; ++k) acc = fma(t[k], v[k], acc);
        e.g += acc;
        op.left(Lc, T1);                                             // E L
        op.left(Gc, EG);                                             // E G
        op.publish(T1);
        TGP_GUNROLL for (int i = 0; i < D; ++i) {
            double a2 = 0.0;
            TGP_GUNROLL for (int k = 0; k < D; ++k) a2 = fma(op.tile[i + G * k], t[k], a2);
            e.L[i] += a2;
            e.E[i] = EG[i];
        }
    }

    // xs <- N(G xs.m + g, G xs.P G' + L)      (predict with (G, g, L)); Xc = row j of G
    __device__ __forceinline__ void smooth_step(double& mj, double* Pc, const double* Gc, const double* Xc, double gj, const double* Lc) const {
        double v[D], T1[D];
        op.gather(mj, v);
        double acc = 0.0;
        TGP_GUNROLL for (int k = 0; k < D; ++k) acc = fma(Xc[k], v[k], acc);
        mj = acc + gj;
        op.publish(Gc);
        op.left(Pc, T1);                                             // G P
        op.publish(T1);
        TGP_GUNROLL for (int i = 0; i < D; ++i) {
            double a2 = 0.0;
            TGP_GUNROLL for (int k = 0; k < D; ++k) a2 = fma(op.tile[i + G * k], Xc[k], a2);
            Pc[i] = a2 + Lc[i];
        }
    }
};

// ---------------------------------------------------------------- pass 2, MODE 2
// MAT == false: MODE 2 (scratch + chunk element). MAT == true: MODE 3 (G_t, g_t, L_t per step to G_out / g_out / L_out; no scratch,
// no element). A compile-time switch: the extra stores change the register allocation of this spill-prone kernel at d = 16.
// LTI == false: the general (per-step) layout -- every time step loads column j of its own A, Q, its a_j and its emission row
// (GroupStep, tgp_group.hpp), A through the group's own LDS tile; no register prefetch here (these kernels are register-bound)
template <int D, bool MAT, bool LTI>
__global__ __launch_bounds__(256) void k_group_apply_posterior(ModelView mv, int L0, int64_t n0, const double* __restrict__ S0,
                                                               double* __restrict__ fs, double* __restrict__ R0, double* __restrict__ partial,
                                                               double* __restrict__ G_out, double* __restrict__ g_out, double* __restrict__ L_out) {
    constexpr int G = GroupGeom<D>::G, NGRP = GroupGeom<D>::NGRP;
    __shared__ __attribute__((aligned(16))) double sA[(LTI ? 1 : NGRP) * G * G];
    __shared__ double tiles[NGRP * GroupGeom<D>::LD];
    __shared__ double sh[12];
    GroupLane<D> gl;
    double Qc[D], H[D], aj, hh, Rsh;
    group_setup<D>(mv, sA, tiles, gl, Qc, H, aj, hh, Rsh);
    if (!LTI) {
        gl.sA = sA + (threadIdx.x / G) * G * G;
        group_clear_A<D>(const_cast<double*>(gl.sA), gl.j);
    }
    const int j = gl.j;
    const int jcl = gl.act ? j : 0;
    GroupRts<D> rts{GroupOps<D>{j, gl.act, gl.tile}, gl.sA};
    const int64_t c = (int64_t)blockIdx.x * NGRP + (threadIdx.x / G);
    int64_t r0, r1;
    chunk_range(mv, c < n0 ? c : n0, L0, r0, r1);
    double Pc[D], mj = 0.0;
    TGP_GUNROLL for (int i = 0; i < D; ++i) Pc[i] = (i == j) ? 1.0 : 0.0;
    if (c < n0 && gl.act) {
        mj = S0[(int64_t)j * n0 + c];
        TGP_GUNROLL for (int i = 0; i < D; ++i) {
            const int lo = i < j ? i : j, hi = i < j ? j : i;
            Pc[i] = S0[(int64_t)(D + hi * (hi + 1) / 2 + lo) * n0 + c];
        }
    }
    double Hj = 0.0;
    TGP_GUNROLL for (int i = 0; i < D; ++i) Hj = (i == j) ? H[i] : Hj;
    GAElem<D> rev;
    GAffineMO<D>::identity(rev, j, gl.act);
    double lml = 0.0, nmiss = 0.0;
    bool ok = true;
    GroupObs<G> ob;
    for (int g = 0; g < L0; g += G) {
        ob.load(mv, c, L0, r0, r1, g, j);
        const int64_t rg = r0 + g;
        const int gend = (int)((r1 - rg) < G ? (r1 - rg) : G);
        double sprod = 1.0, quad = 0.0;
        for (int k = 0; k < gend; ++k) {
            double y, R;
            bool miss;
            const int jj = mv.p == 1 ? 0 : (g + k) % mv.p;
            if (LTI) {
                group_obs_row<D>(mv, jj, j, H, Hj, hh, Rsh);
            } else {
                GroupStep<D> st;
                st.load(mv, rg + k, jcl, gl.act);
                if (st.pred) {
                    group_publish_A<D>(const_cast<double*>(gl.sA), st, j, gl.act);
                    TGP_GUNROLL for (int i = 0; i < D; ++i) Qc[i] = st.Qc[i];
                    aj = st.aj;
                }
                TGP_GUNROLL for (int i = 0; i < D; ++i) H[i] = st.H[i];
                Hj = 0.0;
                TGP_GUNROLL for (int i = 0; i < D; ++i) Hj = (i == j) ? H[i] : Hj;
                hh = st.hh;
                if (mv.p > 1 && mv.sR == 0) Rsh = mv.R[jj];      // shared diagonal noise of a vector observation: row jj (group_setup left R[0])
            }
            ob.step(mv, Rsh, k, y, R, miss);
            if (jj == 0) {   // Forward models only: every time step predicts (at its first observation)
                double Pf[D], Gc[D], Xc[D], Lc[D], gj;
                const double mf = mj;
                TGP_GUNROLL for (int i = 0; i < D; ++i) Pf[i] = Pc[i];
                rts.predict(mj, aj, Pc, Qc);
                ok = rts.invert_dynamics(mf, Pf, mj, Pc, Gc, Xc, gj, Lc) && ok;
                if (MAT) {                    // MODE 3: the time-reversed transition of this step (lgssm.jl:215-221)
                    if (gl.act) {
                        const int64_t te = (rg + k) / mv.p;
                        TGP_GUNROLL for (int i = 0; i < D; ++i) {
                            G_out[te * D * D + i + j * D] = Gc[i];
                            L_out[te * D * D + i + j * D] = Lc[i];
                        }
                        g_out[te * D + j] = gj;
                    }
                } else {
                    rts.extend_right(rev, Gc, gj, Lc);
                }
            }
            double vj = 0.0;
            TGP_GUNROLL for (int i = 0; i < D; ++i) vj = fma(Pc[i], H[i], vj);
            const double S = group_sum<G>(Hj * vj) + R;
            const double hm = group_sum<G>(Hj * mj);
            ok = ok && (S > 0.0);
            const double iS = 1.0 / S;
            const double v = y - (hm + hh);
            const double viS = v * iS;
            mj = fma(vj, viS, mj);
            double V[D];
            gl.gather(vj, V);
            const double wgt = vj * iS;
            TGP_GUNROLL for (int i = 0; i < D; ++i) Pc[i] = fma(-V[i], wgt, Pc[i]);
            quad += v * viS;
            sprod *= S;
            if (sprod > 1e100 || sprod < 1e-100) {
                lml -= 0.5 * log(sprod);
                sprod = 1.0;
            }
            nmiss += miss ? 1.0 : 0.0;
            if (!MAT && gl.act) gfs_store<D>(fs, gfs_index<D>(c, g + k, L0), j, mj, Pc);
        }
        if (gend > 0) lml -= 0.5 * (gend * kLog2Pi + log(sprod) + quad);
    }
    if (!MAT && c < n0 && r1 > r0 && gl.act) GAffineMO<D>::store(rev, R0, n0, n0 - 1 - c, j);
    double a = (j == 0 && c < n0) ? lml : 0.0, b = (j == 0 && c < n0) ? nmiss : 0.0;
    int bad = (j == 0 && c < n0 && !ok) ? 1 : 0;
    block_sum3(a, b, bad, sh);
    if (threadIdx.x == 0) {
        partial[3 * (int64_t)blockIdx.x + 0] = a;
        partial[3 * (int64_t)blockIdx.x + 1] = b;
        partial[3 * (int64_t)blockIdx.x + 2] = (double)bad;
    }
}

// ---------------------------------------------------------------- pass 3
template <int D, bool LTI>
__global__ __launch_bounds__(256) void k_group_smooth(ModelView mv, int L0, int64_t n0, const double* __restrict__ S0, const double* __restrict__ S0r,
                                                      const double* __restrict__ fs, const double* __restrict__ Rnew, int64_t sRn,
                                                      double* __restrict__ mean_out, double* __restrict__ var_out, int* __restrict__ bad,
                                                      GroupAltEmit alt) {
    constexpr int G = GroupGeom<D>::G, NGRP = GroupGeom<D>::NGRP;
    __shared__ __attribute__((aligned(16))) double sA[(LTI ? 1 : NGRP) * G * G];
    __shared__ double tiles[NGRP * GroupGeom<D>::LD];
    GroupLane<D> gl;
    double Qc[D], H[D], aj, hh, Rsh;
    group_setup<D>(mv, sA, tiles, gl, Qc, H, aj, hh, Rsh);
    if (!LTI) {
        gl.sA = sA + (threadIdx.x / G) * G * G;
        group_clear_A<D>(const_cast<double*>(gl.sA), gl.j);      // (before any group leaves: whole groups return together below)
    }
    const int j = gl.j;
    const int jcl = gl.act ? j : 0;
    GroupRts<D> rts{GroupOps<D>{j, gl.act, gl.tile}, gl.sA};
    const int64_t c = (int64_t)blockIdx.x * NGRP + (threadIdx.x / G);
    if (c >= n0) return;                        // whole groups leave together; no block-level barrier below
    int64_t r0, r1;
    chunk_range(mv, c, L0, r0, r1);
    GState<D> xs, carry;
    gstate_load<D>(xs, S0r, n0, n0 - 1 - c, j, gl.act);
    gstate_load<D>(carry, S0, n0, c, j, gl.act);
    double Hj = 0.0;
    TGP_GUNROLL for (int i = 0; i < D; ++i) Hj = (i == j) ? H[i] : Hj;
    bool ok = true;
    for (int64_t r = r1 - 1; r >= r0; --r) {
        const int jj = mv.p == 1 ? 0 : (int)((r - r0) % mv.p);
        if (alt.H != nullptr) {
            // the smoothed state of this time step (its last observation is processed first) through the alternative block
            if (jj == mv.p - 1) {
                const int64_t tt = c * (int64_t)(L0 / mv.p) + (r - r0) / mv.p;
                for (int q = 0; q < alt.pn; ++q) {
                    double pq = 0.0, hq = 0.0;
                    TGP_GUNROLL for (int i = 0; i < D; ++i) {
                        const double hi = alt.H[q * D + i];
                        pq = fma(xs.P[i], hi, pq);
                        hq = (i == j) ? hi : hq;
                    }
                    const double mean = group_sum<G>(hq * xs.m) + alt.h[q];
                    const double var = group_sum<G>(hq * pq);
                    if (j == 0) {
                        mean_out[tt * alt.pn + q] = mean;
                        var_out[tt * alt.pn + q] = var + (alt.sR == 0 ? alt.R[q] : alt.R[tt * alt.pn + q]);
                    }
                }
            }
        } else {
        if (LTI) {
            group_obs_row<D>(mv, jj, j, H, Hj, hh, Rsh);
        } else {
            GroupStep<D> st;          // this step's emission row and (first observation of a time step) its transition
            st.load(mv, r, jcl, gl.act);
            if (st.pred) {
                group_publish_A<D>(const_cast<double*>(gl.sA), st, j, gl.act);
                TGP_GUNROLL for (int i = 0; i < D; ++i) Qc[i] = st.Qc[i];
                aj = st.aj;
            }
            TGP_GUNROLL for (int i = 0; i < D; ++i) H[i] = st.H[i];
            Hj = 0.0;
            TGP_GUNROLL for (int i = 0; i < D; ++i) Hj = (i == j) ? H[i] : Hj;
            hh = st.hh;
            if (mv.p > 1 && mv.sR == 0) Rsh = mv.R[jj];      // shared diagonal noise of a vector observation: row jj (group_setup left R[0])
        }
        // emission marginal of the smoothed state at step r with the NEW noise (lgssm.jl:111-115, missings.jl:35-41)
        double pj = 0.0;
        TGP_GUNROLL for (int i = 0; i < D; ++i) pj = fma(xs.P[i], H[i], pj);
        const double mean = group_sum<G>(Hj * xs.m) + hh;
        const double var = group_sum<G>(Hj * pj);
        const int64_t tm = micro_index(mv, c, (int)(r - r0), L0);
        if (j == 0) {
            mean_out[tm] = mean;
            var_out[tm] = var + (sRn == 0 ? Rnew[jj] : Rnew[tm]);
        }
        }
        if (jj != 0) continue;                  // inside a time step the state does not move
        // filtered state before this step, its prediction, the backward kernel, one RTS step
        double mf, Pf[D];
        if (r == r0) {
            mf = carry.m;
            TGP_GUNROLL for (int i = 0; i < D; ++i) Pf[i] = carry.P[i];
        } else {
            gfs_load<D>(fs, gfs_index<D>(c, (int)(r - r0) - 1, L0), j, gl.act, mf, Pf);
        }
        double mp = mf, Pp[D], Gc[D], Xc[D], Lc[D], gj;
        TGP_GUNROLL for (int i = 0; i < D; ++i) Pp[i] = Pf[i];
        rts.predict(mp, aj, Pp, Qc);
        ok = rts.invert_dynamics(mf, Pf, mp, Pp, Gc, Xc, gj, Lc) && ok;
        rts.smooth_step(xs.m, xs.P, Gc, Xc, gj, Lc);
    }
    if (!ok && j == 0) atomicOr(bad, 1);
}

}  // namespace TGP_NS

// ---------------------------------------------------------------- affine passes in the group layout: prior marginals and rand
// marginals(model) (lgssm.jl:99-115): x <- predict(x) once per time step, N(H x + h, H P H' + R) per observation.
// rand(rng, model) with supplied noise (lgssm.jl:65-91): x <- A x + a + chol(Q + 1e-9 I).U' eps_t (lgc.jl:84-87),
// y = H x + h + sqrt(R [+ 1e-9]) eps_e. Pass 1 composes the chunk's affine element (E, g, L) <- (A E, A g + c, A L A' + Q)
// (rand: c = a + noise, L stays 0); pass 2 propagates the state. Both orderings (the first processed time step of a
// Reverse-ordered model does not predict; GroupStep / trans_index pick the blocks) and both layouts (LTI: blocks from
// group_setup; per-step: GroupStep loads, A through the group's own LDS tile).
namespace TGP_NS {

// upper Cholesky factor of S + jitter I, column j per lane (Uc[i] = U[i][j]); S given by columns (Sc[i] = S[i][j])
template <int D>
__device__ __forceinline__ bool group_chol_upper(const GroupOps<D>& op, const double* Sc, double jitter, double* Uc) {
    constexpr int V0 = GroupGeom<D>::V0, V1 = GroupGeom<D>::V1;
    const int j = op.j;
    bool ok = true;
    TGP_GUNROLL for (int i = 0; i < D; ++i) Uc[i] = 0.0;
    TGP_GUNROLL for (int i = 0; i < D; ++i) {
        double acc = Sc[i] + ((i == j) ? jitter : 0.0);
        wave_sync();
        if (j == i) {
            double dg = acc;
            TGP_GUNROLL for (int k = 0; k < i; ++k) dg = fma(-Uc[k], Uc[k], dg);
            const double di = sqrt(dg);
            TGP_GUNROLL for (int k = 0; k < i; ++k) op.tile[V0 + k] = Uc[k];
            op.tile[V1 + 0] = dg;
            op.tile[V1 + 1] = di;
            op.tile[V1 + 2] = 1.0 / di;
        }
        wave_sync();
        const double dg = op.tile[V1 + 0], di = op.tile[V1 + 1], ri = op.tile[V1 + 2];
        ok = ok && (dg > 0.0);
        TGP_GUNROLL for (int k = 0; k < i; ++k) acc = fma(-op.tile[V0 + k], Uc[k], acc);
        Uc[i] = (j == i) ? di : ((j > i && op.act) ? acc * ri : 0.0);
    }
    return ok;
}

// vj <- sum_k A[j][k] v_k + add   (v distributed one element per lane; A in the LDS tile, row-major [G i + k])
template <int D>
__device__ __forceinline__ void group_affine_mean(const GroupLane<D>& gl, double& vj, double add) {
    constexpr int G = GroupGeom<D>::G;
    double v[D];
    gl.gather(vj, v);
    double acc = 0.0;
    TGP_GUNROLL for (int k = 0; k < D; ++k) acc = fma(gl.sA[G * gl.j + k], v[k], acc);
    vj = acc + add;
}

// (U' eps)_j = sum_{i <= j} U[i][j] eps_i: lane-local with the lane's own column of U
template <int D> __device__ __forceinline__ double group_noise(const double* Uc, const double* __restrict__ ep, int j, bool act) {
    double nz = 0.0;
    TGP_GUNROLL for (int i = 0; i < D; ++i) nz = (i <= j && act) ? fma(Uc[i], ep[i], nz) : nz;
    return nz;
}

template <int D, bool LTI, bool RAND>
__global__ __launch_bounds__(256) void k_group_reduce_marginals(ModelView mv, int L0, int64_t n0, const double* __restrict__ eps_t,
                                                                double* __restrict__ E0, int* __restrict__ bad) {
    constexpr int G = GroupGeom<D>::G, NGRP = GroupGeom<D>::NGRP;
    __shared__ __attribute__((aligned(16))) double sA[(LTI ? 1 : NGRP) * G * G];
    __shared__ double tiles[NGRP * GroupGeom<D>::LD];
    GroupLane<D> gl;
    double Qc[D], H[D], aj, hh, Rsh;
    group_setup<D>(mv, sA, tiles, gl, Qc, H, aj, hh, Rsh);
    if (!LTI) {
        gl.sA = sA + (threadIdx.x / G) * G * G;
        group_clear_A<D>(const_cast<double*>(gl.sA), gl.j);
    }
    const int j = gl.j;
    const int jcl = gl.act ? j : 0;
    const GroupOps<D> op{j, gl.act, gl.tile};
    const int64_t c = (int64_t)blockIdx.x * NGRP + (threadIdx.x / G);
    if (c >= n0) return;
    int64_t r0, r1;
    chunk_range(mv, c, L0, r0, r1);
    GAElem<D> e;
    GAffineMO<D>::identity(e, j, gl.act);
    double Uc[D];
    bool ok = true;
    if (RAND && LTI) ok = group_chol_upper<D>(op, Qc, 1e-9, Uc);
    for (int64_t r = r0; r < r1; r += mv.p) {           // one iteration per time step of the chunk (chunks hold whole time steps)
        const int64_t tproc = r / mv.p;
        bool pred = !(mv.ordering != 0 && tproc == 0);
        if (!LTI) {
            GroupStep<D> st;
            st.load(mv, r, jcl, gl.act);
            pred = st.pred;
            if (pred) {
                group_publish_A<D>(const_cast<double*>(gl.sA), st, j, gl.act);
                TGP_GUNROLL for (int i = 0; i < D; ++i) Qc[i] = st.Qc[i];
                aj = st.aj;
            }
        }
        if (!pred) continue;
        double T1[D];
        gl.mul_A(e.E, T1);
        TGP_GUNROLL for (int i = 0; i < D; ++i) e.E[i] = T1[i];
        if (RAND) {
            if (!LTI) ok = group_chol_upper<D>(op, Qc, 1e-9, Uc) && ok;
            const double* ep = eps_t + trans_index(mv, tproc) * D;
            group_affine_mean<D>(gl, e.g, aj + group_noise<D>(Uc, ep, j, gl.act));
        } else {
            gl.predict(e.g, aj, e.L, Qc);
        }
    }
    if (r1 > r0 && gl.act) GAffineMO<D>::store(e, E0, n0, c, j);
    if (RAND && !ok && j == 0) atomicOr(bad, 1);
}

template <int D, bool LTI, bool RAND>
__global__ __launch_bounds__(256) void k_group_apply_marginals(ModelView mv, int L0, int64_t n0, const double* __restrict__ S0,
                                                               const double* __restrict__ eps_t, const double* __restrict__ eps_e,
                                                               double* __restrict__ mean_out, double* __restrict__ var_out, int* __restrict__ bad) {
    constexpr int G = GroupGeom<D>::G, NGRP = GroupGeom<D>::NGRP;
    __shared__ __attribute__((aligned(16))) double sA[(LTI ? 1 : NGRP) * G * G];
    __shared__ double tiles[NGRP * GroupGeom<D>::LD];
    GroupLane<D> gl;
    double Qc[D], H[D], aj, hh, Rsh;
    group_setup<D>(mv, sA, tiles, gl, Qc, H, aj, hh, Rsh);
    if (!LTI) {
        gl.sA = sA + (threadIdx.x / G) * G * G;
        group_clear_A<D>(const_cast<double*>(gl.sA), gl.j);
    }
    const int j = gl.j;
    const int jcl = gl.act ? j : 0;
    const GroupOps<D> op{j, gl.act, gl.tile};
    const int64_t c = (int64_t)blockIdx.x * NGRP + (threadIdx.x / G);
    if (c >= n0) return;
    int64_t r0, r1;
    chunk_range(mv, c, L0, r0, r1);
    GState<D> x;
    gstate_load<D>(x, S0, n0, c, j, gl.act);
    double Hj = 0.0;
    TGP_GUNROLL for (int i = 0; i < D; ++i) Hj = (i == j) ? H[i] : Hj;
    double Uc[D];
    bool ok = true;
    if (RAND && LTI) ok = group_chol_upper<D>(op, Qc, 1e-9, Uc);
    for (int64_t r = r0; r < r1; ++r) {
        const int64_t tproc = r / mv.p;
        const int jj = (int)(r - tproc * mv.p);
        bool pred = jj == 0 && !(mv.ordering != 0 && tproc == 0);
        if (LTI) {
            group_obs_row<D>(mv, jj, j, H, Hj, hh, Rsh);
        } else {
            GroupStep<D> st;
            st.load(mv, r, jcl, gl.act);
            pred = st.pred;
            if (pred) {
                group_publish_A<D>(const_cast<double*>(gl.sA), st, j, gl.act);
                TGP_GUNROLL for (int i = 0; i < D; ++i) Qc[i] = st.Qc[i];
                aj = st.aj;
            }
            TGP_GUNROLL for (int i = 0; i < D; ++i) H[i] = st.H[i];
            Hj = 0.0;
            TGP_GUNROLL for (int i = 0; i < D; ++i) Hj = (i == j) ? H[i] : Hj;
            hh = st.hh;
            if (mv.p > 1 && mv.sR == 0) Rsh = mv.R[jj];      // shared diagonal noise of a vector observation: row jj (group_setup left R[0])
        }
        if (pred) {
            if (RAND) {
                if (!LTI) ok = group_chol_upper<D>(op, Qc, 1e-9, Uc) && ok;
                const double* ep = eps_t + trans_index(mv, tproc) * D;
                group_affine_mean<D>(gl, x.m, aj + group_noise<D>(Uc, ep, j, gl.act));
            } else {
                gl.predict(x.m, aj, x.P, Qc);
            }
        }
        const int64_t tm = micro_index(mv, c, (int)(r - r0), L0);
        const double Rv = mv.sR != 0 ? mv.R[tm] : Rsh;
        const double mean = group_sum<G>(Hj * x.m) + hh;
        if (RAND) {
            // scalar emission: sqrt(R) eps (lgc.jl:241-243); vector emission with diagonal R: chol(R + 1e-9 I).U' eps (lgc.jl:84-87)
            if (j == 0) mean_out[tm] = mean + sqrt(mv.small_out ? Rv + 1e-9 : Rv) * eps_e[tm];
        } else {
            double pj = 0.0;
            TGP_GUNROLL for (int i = 0; i < D; ++i) pj = fma(x.P[i], H[i], pj);
            const double var = group_sum<G>(Hj * pj);
            if (j == 0) {
                mean_out[tm] = mean;
                var_out[tm] = var + Rv;
            }
        }
    }
    if (RAND && !ok && j == 0) atomicOr(bad, 1);
}

}  // namespace TGP_NS
