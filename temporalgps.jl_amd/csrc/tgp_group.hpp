// Group-per-chunk kernels for state dimensions 5..8 (LTI family, scalar observations): EIGHT lanes share one chunk, lane j
// holding column j of every matrix, so a lane carries ~10 d doubles instead of the 3 d^2 of the lane-per-chunk kernels --
// no spills, full occupancy. (tgp_chunk_body.inc keeps every other case; the scan stages are shared: the element /
// carry-state formats in HBM are the interface.)
//
// Building blocks, all inside one wave (a group never spans waves, so `wave_sync` is the only synchronisation):
//   A X        lane-local: Y[:, j] = A X[:, j]; A (shared, LTI) is read from LDS as a broadcast
//   A P A'     = A (A P)' for symmetric P: ONE transpose of W = A P through the group's LDS tile (lane j writes column j,
//              reads row j), then column j of the result is A (row j of W)'
//   P H        lane-local by symmetry: (P H)_j = H . P[:, j]
//   H'x, sums  3-step xor butterfly over the 8 lanes (every lane ends with the same value)
//   all-gather each lane contributes one value, all read the 8 of them back from the tile
// Same recursions as predict / update_scalar_nolog / f_extend (tgp_math_body.inc), same operation order inside each
// inner product; the results agree with the lane-per-chunk kernels to rounding.
#pragma once

namespace TGP_NS {

// Geometry: G = 8 lanes per chunk for d <= 8, 16 for d <= 16. A group's LDS tile is a G x G matrix (i + G col), two
// G-vectors and 2 doubles of padding (82 / 290 doubles: the groups of a wave start 4 banks apart).
template <int D> struct GroupGeom {
    static_assert(D >= 1 && D <= 16, "group kernels: d <= 16");
    static constexpr int G = D <= 8 ? 8 : 16;
    static constexpr int LOG2G = D <= 8 ? 3 : 4;
    static constexpr int NGRP = 256 / G;           // groups (chunks) per 256-thread block
    static constexpr int V0 = G * G, V1 = G * G + G;
    static constexpr int LD = G * G + 2 * G + 2;
};
#define TGP_GUNROLL _Pragma("unroll")              // the group code keeps its (short) arrays in registers for every d

// Sum over the G lanes of a group, every lane ending with the same value. Data-parallel primitives instead of
// ds_bpermute: quad_perm [1,0,3,2] and [2,3,0,1] inside quads, row_half_mirror across the two quads of 8 lanes, row_mirror
// across the two halves of 16 (a + b == b + a bit for bit, so all lanes agree).
template <int CTRL> __device__ __forceinline__ double dpp_move(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
template <int G> __device__ __forceinline__ double group_sum(double x) {
    x += dpp_move<0xB1>(x);                 // quad_perm:[1,0,3,2]
    x += dpp_move<0x4E>(x);                 // quad_perm:[2,3,0,1]
    x += dpp_move<0x141>(x);                // row_half_mirror
    if (G == 16) x += dpp_move<0x140>(x);   // row_mirror
    return x;
}

template <int D> struct GroupLane {
    static constexpr int G = GroupGeom<D>::G, V0 = GroupGeom<D>::V0, V1 = GroupGeom<D>::V1;
    int j;                   // lane inside the group == matrix column it owns
    double* tile;            // the group's LDS tile: [0, 64) an 8 x 8 matrix (i + 8 col), [64, 72) and [72, 80) two vectors
    const double* sA;        // A in LDS, row-major [8 i + k] (a row is read with 16-byte loads, broadcast to the wave)
    bool act;                // j < D
    // y[:, j] = A x[:, j]
    __device__ __forceinline__ void mul_A(const double* x, double* y) const {
        TGP_GUNROLL for (int i = 0; i < D; ++i) {
            double acc = 0.0;
            TGP_GUNROLL for (int k = 0; k < D; ++k) acc = fma(sA[G * i + k], x[k], acc);
            y[i] = acc;
        }
    }
    // every lane gets all D elements of a vector distributed one element per lane
    __device__ __forceinline__ void gather(double vj, double* v) const {
        wave_sync();
        tile[V0 + j] = vj;
        wave_sync();
        TGP_GUNROLL for (int k = 0; k < D; ++k) v[k] = tile[V0 + k];
    }
    __device__ __forceinline__ void gather2(double vj, double wj, double* v, double* w) const {
        wave_sync();
        tile[V0 + j] = vj;
        tile[V1 + j] = wj;
        wave_sync();
        TGP_GUNROLL for (int k = 0; k < D; ++k) { v[k] = tile[V0 + k]; w[k] = tile[V1 + k]; }
    }
    // One exchange for a whole predict:  v <- A v + a  (element j of v in lane j)  and  S <- A S A' + Q  (symmetric S by
    // columns): W = A S is lane-local, its transpose and the gather of v share ONE trip through the tile, then column j of
    // A S A' is A (row j of W)'.
    __device__ __forceinline__ void predict(double& vj, double aj, double* Sc, const double* Qc) const {
        double W[D], row[D], v[D];
        mul_A(Sc, W);
        wave_sync();
        TGP_GUNROLL for (int i = 0; i < D; ++i) tile[i + G * j] = W[i];
        tile[V0 + j] = vj;
        wave_sync();
        TGP_GUNROLL for (int k = 0; k < D; ++k) {
            row[k] = act ? tile[j + G * k] : 0.0;      // rows >= D of the tile are never written
            v[k] = tile[V0 + k];
        }
        mul_A(row, Sc);
        TGP_GUNROLL for (int i = 0; i < D; ++i) Sc[i] += Qc[i];
        double acc = 0.0;
        TGP_GUNROLL for (int k = 0; k < D; ++k) acc = fma(sA[G * j + k], v[k], acc);
        vj = acc + aj;
    }
};

// observation stream of a group: 8 consecutive processing steps are loaded by the 8 lanes (one coalesced 64-byte row
// per group) and handed out step by step with a shuffle
template <int G> struct GroupObs {
    double yv, rv;
    int mv_;
    __device__ __forceinline__ void load(const ModelView& mv, int64_t c, int L0, int64_t r0, int64_t r1, int g, int j) {
        const int64_t r = r0 + g + j;
        yv = 0.0;
        rv = 0.0;
        mv_ = 0;
        if (r < r1) {
            const int64_t tm = micro_index(mv, c, g + j, L0);
            yv = mv.y[tm];
            if (mv.sR != 0) rv = mv.R[tm];
            if (mv.missing != nullptr) mv_ = mv.missing[tm];
        }
    }
    __device__ __forceinline__ void step(const ModelView& mv, double Rshared, int k, double& y, double& R, bool& miss) const {
        y = __shfl(yv, k, G);
        R = mv.sR != 0 ? __shfl(rv, k, G) : Rshared;
        miss = __shfl(mv_, k, G) != 0;
        if (miss) { y = 0.0; R = kLargeVar; }
    }
};

template <int D> __device__ __forceinline__ void group_setup(const ModelView& mv, double* sA, double* tiles, GroupLane<D>& gl, double* Qc,
                                                             double* H, double& aj, double& hh, double& Rsh) {
    constexpr int G = GroupGeom<D>::G;
    const int tid = threadIdx.x;
    if (tid < G * G) {
        const int i = tid / G, k = tid % G;
        sA[tid] = (i < D && k < D) ? mv.A[i + k * D] : 0.0;      // (per-step layouts: the first block; overwritten per step)
    }
    __syncthreads();
    gl.j = tid & (G - 1);
    gl.act = gl.j < D;
    gl.tile = tiles + (tid / G) * GroupGeom<D>::LD;
    gl.sA = sA;
    const int jc = gl.act ? gl.j : 0;
    TGP_GUNROLL for (int i = 0; i < D; ++i) { Qc[i] = gl.act ? mv.Q[i + jc * D] : 0.0; H[i] = mv.H[i]; }
    aj = gl.act ? mv.a[jc] : 0.0;
    hh = mv.h[0];
    Rsh = mv.sR == 0 ? mv.R[0] : 0.0;
}

// Vector observations (p > 1, shared emission block H [p][d], h [p], R [p]): processing step i of a chunk is observation
// row i % p of its time step (chunks hold whole time steps); the row is re-read (wave-uniform, cached) per step.
template <int D> __device__ __forceinline__ void group_obs_row(const ModelView& mv, int jj, int j, double* H, double& Hj, double& hh, double& Rsh) {
    if (mv.p == 1) return;
    TGP_GUNROLL for (int i = 0; i < D; ++i) H[i] = mv.H[jj * D + i];
    Hj = 0.0;
    TGP_GUNROLL for (int i = 0; i < D; ++i) Hj = (i == j) ? H[i] : Hj;
    hh = mv.h[jj];
    Rsh = mv.sR == 0 ? mv.R[jj] : 0.0;
}

// ---------------------------------------------------------------- general (per-step) layout in the group kernels
// Every time step carries its own A, a, Q and / or H, h (what irregular spacing and prediction at new inputs produce,
// lti_sde.jl:135-146; the strides in ModelView say which). Lane j needs only COLUMN j of A and Q, a_j and the emission row, so a
// step's record costs ~3 d + 2 registers per lane (the lane-per-chunk kernels hold 2 d^2 + ... per lane and spill from d = 6).
// The record of step k+1 is loaded while step k is computed (one step of software prefetch: its HBM round trip would
// otherwise head every iteration of a sequential loop); A goes through the group's own LDS tile (`sA`, row-major), where
// GroupLane::mul_A reads it as a broadcast. The arrays are read in the reference layout (ModelView::A .. h + stride): a
// group's 8 lanes touch one contiguous d x d block per array and step.
template <int D> struct GroupStep {
    double Ac[D], Qc[D], H[D], aj, hh;
    bool pred;
    // transition of processing time step `tproc` (column jc of A and Q, element jc of a) and emission row jj of the same step
    __device__ __forceinline__ void load(const ModelView& mv, int64_t r, int jc, bool act) {
        const int64_t tproc = r / mv.p;
        const int jj = (int)(r - tproc * mv.p);
        pred = jj == 0 && !(mv.ordering != 0 && tproc == 0);
        const int64_t tt = trans_index(mv, tproc), te = time_index(mv, tproc);
        if (pred) {
            const double* Ap = mv.A + tt * mv.sA + jc * D;
            const double* Qp = mv.Q + tt * mv.sQ + jc * D;
            TGP_GUNROLL for (int i = 0; i < D; ++i) { Ac[i] = act ? Ap[i] : 0.0; Qc[i] = act ? Qp[i] : 0.0; }
            aj = act ? mv.a[tt * mv.sa + jc] : 0.0;
        }
        const double* Hp = mv.H + te * mv.sH + jj * D;
        TGP_GUNROLL for (int i = 0; i < D; ++i) H[i] = Hp[i];
        hh = mv.h[te * mv.sh + jj];
    }
};

// The group's own A tile starts as zeros: group_publish_A below only ever writes rows and columns < D, but lane j >= D (inactive,
// G > D) reads row j of the tile for its (discarded) mean element, and 0 * (a NaN left in LDS by an earlier kernel) would poison
// the group sums. (Found in round 2: the run-time check rejected the per-step group kernels or not depending on what had run
// before in the process.)
template <int D> __device__ __forceinline__ void group_clear_A(double* sAg, int j) {
    constexpr int G = GroupGeom<D>::G;
    TGP_GUNROLL for (int i = 0; i < G; ++i) sAg[G * i + j] = 0.0;
    wave_sync();
}

// publish the step's A in the group's LDS tile (row-major [G i + k]): lane k owns column k
template <int D> __device__ __forceinline__ void group_publish_A(double* sAg, const GroupStep<D>& st, int j, bool act) {
    constexpr int G = GroupGeom<D>::G;
    wave_sync();
    if (act) {
        TGP_GUNROLL for (int i = 0; i < D; ++i) sAg[G * i + j] = st.Ac[i];
    }
    wave_sync();
}

// ---------------------------------------------------------------- pass 1: the chunk's filter element
template <int D, bool LTI>
__global__ __launch_bounds__(256) void k_group_reduce_filter(ModelView mv, int L0, int64_t n0, double* __restrict__ E0) {
    constexpr int G = GroupGeom<D>::G, NGRP = GroupGeom<D>::NGRP;
    __shared__ __attribute__((aligned(16))) double sA[(LTI ? 1 : NGRP) * G * G];
    __shared__ double tiles[NGRP * GroupGeom<D>::LD];
    GroupLane<D> gl;
    double Qc[D], H[D], aj, hh, Rsh;
    group_setup<D>(mv, sA, tiles, gl, Qc, H, aj, hh, Rsh);
    if (!LTI) {
        gl.sA = sA + (threadIdx.x / G) * G * G;      // the group's own A tile (rewritten per step)
        group_clear_A<D>(const_cast<double*>(gl.sA), gl.j);
    }
    const int j = gl.j;
    const int64_t c = (int64_t)blockIdx.x * NGRP + (threadIdx.x / G);
    int64_t r0, r1;
    chunk_range(mv, c < n0 ? c : n0, L0, r0, r1);
    // element: identity
    double Ac[D], Cc[D], Jc[D], bj = 0.0, etaj = 0.0;
    TGP_GUNROLL for (int i = 0; i < D; ++i) { Ac[i] = (i == j) ? 1.0 : 0.0; Cc[i] = 0.0; Jc[i] = 0.0; }
    double Hj = 0.0;
    TGP_GUNROLL for (int i = 0; i < D; ++i) Hj = (i == j) ? H[i] : Hj;
    GroupObs<G> ob;
    GroupStep<D> nxt;
    constexpr bool PF = D <= 8;       // one step of register prefetch; d >= 9 (spill-bound already) loads the step when it is used
    const int jcl = gl.act ? j : 0;
    if (!LTI && PF && r0 < r1) nxt.load(mv, r0, jcl, gl.act);
    for (int g = 0; g < L0; g += G) {
        ob.load(mv, c, L0, r0, r1, g, j);
        const int64_t rg = r0 + g;
        const int gend = (int)((r1 - rg) < G ? (r1 - rg) : G);
        for (int k = 0; k < gend; ++k) {
            double y, R;
            bool miss;
            const int jj = mv.p == 1 ? 0 : (g + k) % mv.p;
            bool do_predict;
            if (LTI) {
                group_obs_row<D>(mv, jj, j, H, Hj, hh, Rsh);
                do_predict = jj == 0 && !(mv.ordering != 0 && (rg + k) == 0);
            } else {
                if (!PF) nxt.load(mv, rg + k, jcl, gl.act);
                const GroupStep<D>& cur = nxt;
                GroupStep<D> held;
                if (PF) {
                    held = nxt;
                    if (rg + k + 1 < r1) nxt.load(mv, rg + k + 1, jcl, gl.act);      // in flight while this step is computed
                }
                const GroupStep<D>& use = PF ? held : cur;
                do_predict = use.pred;
                if (do_predict) {
                    group_publish_A<D>(const_cast<double*>(gl.sA), use, j, gl.act);
                    TGP_GUNROLL for (int i = 0; i < D; ++i) Qc[i] = use.Qc[i];
                    aj = use.aj;
                }
                TGP_GUNROLL for (int i = 0; i < D; ++i) H[i] = use.H[i];
                Hj = 0.0;
                TGP_GUNROLL for (int i = 0; i < D; ++i) Hj = (i == j) ? H[i] : Hj;
                hh = use.hh;
                if (mv.p > 1 && mv.sR == 0) Rsh = mv.R[jj];      // shared diagonal noise of a vector observation: row jj (group_setup left R[0])
            }
            ob.step(mv, Rsh, k, y, R, miss);
            if (do_predict) {
                double T1[D];
                gl.mul_A(Ac, T1);                       // Abar <- A Abar
                TGP_GUNROLL for (int i = 0; i < D; ++i) Ac[i] = T1[i];
                gl.predict(bj, aj, Cc, Qc);             // b <- A b + a ; C <- A C A' + Q
            }
            double wj = 0.0, cvj = 0.0;                 // w = Abar' H, Cv = C H
            TGP_GUNROLL for (int i = 0; i < D; ++i) { wj = fma(Ac[i], H[i], wj); cvj = fma(Cc[i], H[i], cvj); }
            const double s = R + group_sum<G>(Hj * cvj);
            const double r = (y - hh) - group_sum<G>(Hj * bj);
            const double is = 1.0 / s;
            etaj = fma(wj, r * is, etaj);
            bj = fma(cvj, r * is, bj);
            double w[D], Cv[D];
            gl.gather2(wj, cvj, w, Cv);
            TGP_GUNROLL for (int i = 0; i < D; ++i) {
                Jc[i] = fma(w[i] * is, wj, Jc[i]);
                Ac[i] = fma(-Cv[i] * is, wj, Ac[i]);
                Cc[i] = fma(-Cv[i] * is, cvj, Cc[i]);
            }
        }
    }
    if (c < n0 && r1 > r0 && gl.act) {
        constexpr int DD = D * D, DS = Dim<D>::DS;
        TGP_GUNROLL for (int i = 0; i < D; ++i) E0[(int64_t)(i + j * D) * n0 + c] = Ac[i];
        E0[(int64_t)(DD + j) * n0 + c] = bj;
        E0[(int64_t)(DD + D + DS + j) * n0 + c] = etaj;
        TGP_GUNROLL for (int i = 0; i < D; ++i)
            if (i <= j) {
                E0[(int64_t)(DD + D + j * (j + 1) / 2 + i) * n0 + c] = Cc[i];
                E0[(int64_t)(DD + 2 * D + DS + j * (j + 1) / 2 + i) * n0 + c] = Jc[i];
            }
    }
}

// ---------------------------------------------------------------- pass 2, logpdf: filter from the chunk's carry-in state
// OUT == true: MODE 1, the filtering distributions are written as well (compile-time: see k_group_apply_posterior)
template <int D, bool OUT, bool LTI>
__global__ __launch_bounds__(256) void k_group_apply_logpdf(ModelView mv, int L0, int64_t n0, const double* __restrict__ S0,
                                                            double* __restrict__ partial, double* __restrict__ m_out,
                                                            double* __restrict__ P_out) {
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
    double lml = 0.0, nmiss = 0.0;
    bool ok = true;
    GroupObs<G> ob;
    GroupStep<D> nxt;
    constexpr bool PF = D <= 8;       // one step of register prefetch; d >= 9 (spill-bound already) loads the step when it is used
    const int jcl = gl.act ? j : 0;
    if (!LTI && PF && r0 < r1) nxt.load(mv, r0, jcl, gl.act);
    for (int g = 0; g < L0; g += G) {
        ob.load(mv, c, L0, r0, r1, g, j);
        const int64_t rg = r0 + g;
        const int gend = (int)((r1 - rg) < G ? (r1 - rg) : G);
        double sprod = 1.0, quad = 0.0;
        for (int k = 0; k < gend; ++k) {
            double y, R;
            bool miss;
            const int jj = mv.p == 1 ? 0 : (g + k) % mv.p;
            bool do_predict;
            if (LTI) {
                group_obs_row<D>(mv, jj, j, H, Hj, hh, Rsh);
                do_predict = jj == 0 && !(mv.ordering != 0 && (rg + k) == 0);
            } else {
                if (!PF) nxt.load(mv, rg + k, jcl, gl.act);
                const GroupStep<D>& cur = nxt;
                GroupStep<D> held;
                if (PF) {
                    held = nxt;
                    if (rg + k + 1 < r1) nxt.load(mv, rg + k + 1, jcl, gl.act);
                }
                const GroupStep<D>& use = PF ? held : cur;
                do_predict = use.pred;
                if (do_predict) {
                    group_publish_A<D>(const_cast<double*>(gl.sA), use, j, gl.act);
                    TGP_GUNROLL for (int i = 0; i < D; ++i) Qc[i] = use.Qc[i];
                    aj = use.aj;
                }
                TGP_GUNROLL for (int i = 0; i < D; ++i) H[i] = use.H[i];
                Hj = 0.0;
                TGP_GUNROLL for (int i = 0; i < D; ++i) Hj = (i == j) ? H[i] : Hj;
                hh = use.hh;
                if (mv.p > 1 && mv.sR == 0) Rsh = mv.R[jj];      // shared diagonal noise of a vector observation: row jj (group_setup left R[0])
            }
            ob.step(mv, Rsh, k, y, R, miss);
            if (do_predict) {
                gl.predict(mj, aj, Pc, Qc);             // m <- A m + a ; P <- A P A' + Q
            }
            double vj = 0.0;                            // V = P H
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
            if (OUT && jj == mv.p - 1 && gl.act) {                  // MODE 1: filtering distribution of this time step
                const int64_t te = time_index(mv, (rg + k) / mv.p);
                m_out[te * D + j] = mj;
                TGP_GUNROLL for (int i = 0; i < D; ++i) P_out[te * D * D + i + j * D] = Pc[i];
            }
        }
        if (gend > 0) lml -= 0.5 * (gend * kLog2Pi + log(sprod) + quad);
    }
    // every lane of a group holds the same (lml, nmiss, ok): lane 0 of each group contributes
    double a = (j == 0 && c < n0) ? lml : 0.0, b = (j == 0 && c < n0) ? nmiss : 0.0;
    int bad = (j == 0 && c < n0 && !ok) ? 1 : 0;
    block_sum3(a, b, bad, sh);
    if (threadIdx.x == 0) {
        partial[3 * (int64_t)blockIdx.x + 0] = a;
        partial[3 * (int64_t)blockIdx.x + 1] = b;
        partial[3 * (int64_t)blockIdx.x + 2] = (double)bad;
    }
}

}  // namespace TGP_NS
