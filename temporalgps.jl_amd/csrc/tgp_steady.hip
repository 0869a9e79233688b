// Stationary-gain scan engine: see tgp_steady.hpp for what it computes and why.  gfx950 only (wave64, LDS as the lane exchange of the
// one-wave setup kernel, __shfl for the in-tile scans).
#include "tgp_steady.hpp"
#include "tgp_alloc.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace tgp_steady {

namespace {

constexpr double kLog2Pi = 1.8378770664093454835606594728112;
constexpr double kTol = 4.5e-16;   // "no longer changes": 2 ulp relative to the element's natural scale
constexpr int kCovScanMaxD = 4;    // filter_cov_scan / filter_cov_reg (matrices in one lane's registers) up to here, filter_cov_lds beyond

// ---- layout of the steady-coefficient block (doubles) ---------------------------------------------------------------------------
template <int D>
struct SS {
    static constexpr int A = 0;             // [D][D] row-major
    static constexpr int a = A + D * D;     // [D]
    static constexpr int h = a + D;         // [D]
    static constexpr int hh = h + D;
    static constexpr int kA = hh + 1;       // [D]  A K
    static constexpr int rS = kA + D;       // R / S
    static constexpr int iS = rS + 1;       // 1 / S
    static constexpr int logS = iS + 1;
    static constexpr int G = logS + 1;      // [D][D] row-major: smoother gain (x_{t-1} = G x_t + g)
    static constexpr int c = G + D * D;     // [D]  G K
    static constexpr int vb = c + D;        // H Ps H'  (smoothed variance without Rnew)
    static constexpr int mu0 = vb + 1;      // [D]  A x0.m + a
    static constexpr int LS = mu0 + D;      // sum of log S_t over the n0 head steps
    static constexpr int size = LS + 1;
};

constexpr int kPowN = 26;          // powers Phi^(2^k), G^(2^k), k < kPowN (the scans need k <= 23, whatever T)
constexpr int kBlk = 8;            // tiles (waves) per workgroup of the tile passes: one carry element per workgroup, 4096 steps
constexpr int kLogTile = 9, kLogBlk = 12;      // kTile = 2^9, kBlk kTile = 2^12
constexpr int kBlkThreads = kBlk * 64;

// ---- the constant block: everything the tile passes read through SCALAR loads (written by k_setup only) -------------------------------
template <int D>
struct CL {
    static constexpr int DD = D * D;
    static constexpr int ss = 0;
    static constexpr int pphi = (SS<D>::size + 7) & ~7;      // [kPowN][DD] row-major: (A - kA h')^(2^k)
    static constexpr int pg = pphi + kPowN * DD;             // [kPowN][DD]: G^(2^k)
    static constexpr int B512 = pg + kPowN * DD;             // coupling of a full tile: d(lam at tile start) / d(mu at tile start) = -B
    static constexpr int Blt = B512 + DD;                    // ... of the LAST tile (nv <= 512 valid steps)
    static constexpr int Bblk = Blt + DD;                   // ... of a full workgroup (kBlk tiles)
    static constexpr int Blb = Bblk + DD;                   // ... of the last workgroup
    static constexpr int PT = Blb + DD;                      // [kBlk][DD]: Phi^(512 w), w = 0..kBlk-1   (tile carries inside a workgroup,
    static constexpr int GT = PT + kBlk * DD;                // [kBlk][DD]: G^(512 w)                      k_apply)
    static constexpr int KW = GT + kBlk * DD;                // [kBlk][DD]: K_w = G^512 K_{w+1} + B_512 Phi^(512 (w+1)), K_{kBlk-1} = 0
    // time shards (tgp_multi): the whole run of stationary steps of THIS segment (L = T - head steps) as one element
    static constexpr int PSeg = KW + kBlk * DD;              // Phi^L
    static constexpr int GSeg = PSeg + DD;                   // G^L
    static constexpr int BSeg = GSeg + DD;                   // B_L
    static constexpr int GLb = BSeg + DD;                    // G^(steps of the ragged last workgroup): the lam a shard receives enters there
    static constexpr int WJ = GLb + DD;                      // [2 kSub][D]: h' Phi^j -- the innovation j steps behind a lane's start state st is r0_j - WJ[j] . st
    static constexpr int size = WJ + 2 * kSub * D;             //   (pass 2 uses the first eight rows, pass 1 -- sixteen steps per lane -- all of them)
};
// slot a time shard hands to the exchange: [0] applies, then F (mu behind the segment under a zero carry-in; rank 0: the mu itself),
// B0 (lam in front of the segment's stationary steps under zero carries), Phi^L, G^L, B_L
template <int D>
struct ShardSlot {
    static constexpr int DD = D * D;
    static constexpr int F = 1, B0 = 1 + D, P = 1 + 2 * D, G = P + DD, B = G + DD, size = B + DD;
};

// record of an adjoint call (doubles): the sums of the stationary tiles, then what the host's half needs (k_final_grad)
template <int D>
struct GradRec {
    static constexpr int DD = D * D;
    static constexpr int SA = 0, Sa = DD, Sk = DD + D, Srm = DD + 2 * D, Sr = DD + 3 * D, SSQ = DD + 3 * D + 1;
    static constexpr int NS = DD + 3 * D + 2;
    static constexpr int psi = NS, mu = NS + D, meta = NS + 2 * D, model = meta + 4;
    static constexpr int size = model + 2 * DD + 2 * D + 2 + D + D * (D + 1) / 2;
};

struct Tab {
    long long* hdr;      // [0] applies (1/0), [1] th (head tiles), [2] n0, [3] n1, [4] not PD
    double* cst;         // CL<D>
    double* ssc;         // = cst + CL::ss
    double* pw_phi;      // = cst + CL::pphi
    double* pw_g;        // = cst + CL::pg
    double *Fb, *B0b;    // [nblk] workgroup elements (zero carries)
    double *MUb, *LAMb;  // [nblk + 1] carries of the workgroups: mu at the first step; lam after the first step has been absorbed
    // head tables: nhmax = kHeadMaxTiles * kTile entries
    double *h_kA, *h_rS, *h_iS, *h_G, *h_c, *h_vb, *h_r;
    double *s_Pf, *s_Pp, *s_L, *s_Ps;    // [nhmax + 1][D*D] scratch of the setup kernel
    double* t_vb;        // [kTailMax]: H Ps H' at step T-1-j
    double* t_Ps;        // [kTailMax][D*D]
    double *F, *B0;      // [ntiles][D] tile elements (zero carries)
    double *Pw, *Lw;     // [ntiles][D] tile carries when the WORKGROUP's carries are zero (mu at the tile's first step; lam behind its last)
    double* SSQ;         // [nblk] sum r^2 of a workgroup's tiles
    double* misc;        // [0] sum r^2 / S over the head; [8..] cycle stamps of k_setup's phases
    double *GS, *grec;   // adjoint calls: [nblk][NS] sums of the workgroups; the call's record (GradRec<D>)
};

// workgroups of the stationary tiles: the launch is sized for one head tile (ntiles1 = ntiles - 1 tiles), the device knows th
__device__ __forceinline__ long long nblk_max_for(long long th, long long ntiles) { return (ntiles - th + kBlk - 1) / kBlk; }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }

// =================================================================================================================================
// k_setup: one wave, lane (i, j) owns element (i, j) of every d x d matrix; LDS is the exchange.  Sequential in time, but only over the
// head (n0 steps), the tail (n1 steps) and log2(T) squarings.
// =================================================================================================================================

// reverse-time dynamics of one step, one lane: lgssm.jl:231-238 with Pf = filtered covariance before the step, Pp = predicted.
template <int D>
__device__ bool invert_dynamics_lane(const double* __restrict__ A /*col-major*/, const double (&Pf)[D][D], const double (&Pp)[D][D],
                                     double (&G)[D][D], double (&L)[D][D]) {
    double U[D][D];
    bool ok = true;
    // U'U = Symmetric(Pp) + 1e-10 I  (upper triangle of Pp)
#pragma unroll
    for (int i = 0; i < D; ++i) {
        double s = Pp[i][i] + 1e-10;
#pragma unroll
        for (int k = 0; k < i; ++k) s -= U[k][i] * U[k][i];
        ok = ok && (s > 0.0);
        const double u = sqrt(s);
        U[i][i] = u;
        const double ru = 1.0 / u;
#pragma unroll
        for (int j = i + 1; j < D; ++j) {
            double v = Pp[i][j];
#pragma unroll
            for (int k = 0; k < i; ++k) v -= U[k][i] * U[k][j];
            U[i][j] = v * ru;
        }
#pragma unroll
        for (int j = 0; j < i; ++j) U[i][j] = 0.0;
    }
    // M = A * Pf (full Pf, as the reference), X = U' \ M, Gt = U \ X
    double X[D][D];
#pragma unroll
    for (int c = 0; c < D; ++c) {
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) v = fma(A[i + k * D], Pf[k][c], v);
#pragma unroll
            for (int k = 0; k < i; ++k) v -= U[k][i] * X[k][c];
            X[i][c] = v / U[i][i];
        }
#pragma unroll
        for (int i = D - 1; i >= 0; --i) {
            double v = X[i][c];
#pragma unroll
            for (int k = i + 1; k < D; ++k) v -= U[i][k] * X[k][c];
            X[i][c] = v / U[i][i];          // X now holds Gt
        }
    }
    // G = Gt', L = Pf - (U Gt)'(U Gt)
    double W[D][D];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int c = 0; c < D; ++c) {
            double v = 0.0;
#pragma unroll
            for (int k = i; k < D; ++k) v = fma(U[i][k], X[k][c], v);
            W[i][c] = v;
        }
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) {
            G[i][j] = X[j][i];
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) v = fma(W[k][i], W[k][j], v);
            L[i][j] = Pf[i][j] - v;
        }
    return ok;
}

// LDS exchange inside ONE wave: DS instructions of a wave execute in order, so a write is visible to the reads that follow it in
// program order; all that is needed is that the compiler keeps that order (no s_barrier, and -- unlike __syncthreads -- no wait for the
// global stores that are in flight: the tables are written fire-and-forget from inside the sequential loops).
// one wave talking to itself through GLOBAL memory as well (what a single-wave workgroup used __syncthreads for)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- phase (a) + (b), d <= 3: everything in registers, every lane runs the same covariance recursion (no exchange, no memory in the
// dependent chain); lane (t mod 64) keeps step t's (Pf, Pp), and after every 64 steps the lanes turn them into that step's reverse-time
// gains (invert_dynamics_lane) in parallel.  Writes the head tables; returns n0 (-1: not settled), the stationary step's values.
template <int D>
__device__ __forceinline__ int filter_cov_reg(const ModelDev& m, const Tab& tb, int lane, bool& bad, double (&Pss)[D][D], double (&kAss)[D],
                                              double& Sss, double& LS) {
    constexpr int DD = D * D;
    constexpr int nhmax = kHeadMaxTiles * kTile;
    double A[D][D], Q[D][D], hv[D], P[D][D], Pold2[D][D], cPf[D][D], cPp[D][D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        hv[i] = m.H[i];
#pragma unroll
        for (int k = 0; k < D; ++k) {
            A[i][k] = m.A[i + k * D];
            Q[i][k] = m.Q[i + k * D];
            const int r = i < k ? i : k, c = i < k ? k : i;
            P[i][k] = m.x0[D + c * (c + 1) / 2 + r];
            Pold2[i][k] = 0.0;
            cPf[i][k] = cPp[i][k] = 0.0;
        }
    }
    const double R = m.R[0];
    double* ss = tb.ssc;
    auto gains = [&](int t) {        // phase (b) for the step this lane has kept
        double G[D][D], L[D][D];
        const bool ok = invert_dynamics_lane<D>(m.A, cPf, cPp, G, L);
        if (!ok) tb.hdr[4] = 1;
        double K[D], Sv = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            double v = 0.0;
#pragma unroll
            for (int l = 0; l < D; ++l) v = fma(hv[l], cPp[l][k], v);
            K[k] = v;
            Sv = fma(v, hv[k], Sv);
        }
        Sv += R;
        const double iSv = 1.0 / Sv;
#pragma unroll
        for (int r = 0; r < D; ++r) {
            double v = 0.0;
#pragma unroll
            for (int c = 0; c < D; ++c) {
                v = fma(G[r][c], K[c] * iSv, v);
                if (t < nhmax) tb.h_G[(size_t)t * DD + r * D + c] = G[r][c];
                tb.s_L[(size_t)t * DD + r * D + c] = L[r][c];
            }
            if (t < nhmax) tb.h_c[t * D + r] = v;
        }
    };
    int tc = -1, n0 = -1;
    double prod = 1.0;
    LS = 0.0;
    for (int t = 0; t <= nhmax; ++t) {
        double t1[D][D], pp[D][D], V[D];
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) {
                double v = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) v = fma(A[i][k], (k <= j ? P[k][j] : P[j][k]), v);       // A * Symmetric(P)
                t1[i][j] = v;
            }
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) {
                double v = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) v = fma(t1[i][k], A[j][k], v);
                pp[i][j] = v + Q[i][j];
            }
        double S = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            double v = 0.0;
#pragma unroll
            for (int l = 0; l < D; ++l) v = fma(hv[l], pp[l][k], v);
            V[k] = v;
            S = fma(v, hv[k], S);
        }
        S += R;
        bad = bad || !(S > 0.0);
        const double iS = 1.0 / S, rs = 1.0 / sqrt(S);
        const bool mine = (t & 63) == lane;
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) v = fma(A[i][k], V[k] * iS, v);
            kAss[i] = v;
            if (mine && t < nhmax) tb.h_kA[t * D + i] = v;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                cPf[i][j] = mine ? P[i][j] : cPf[i][j];
                cPp[i][j] = mine ? pp[i][j] : cPp[i][j];
            }
        }
        if (mine && t < nhmax) {
            tb.h_rS[t] = R * iS;
            tb.h_iS[t] = iS;
        }
        Sss = S;
        if (tc >= 0) {           // this was the extra iteration from the settled covariance: the stationary step
            n0 = t;
            break;
        }
        if (t == nhmax) break;
        prod *= S;
        if ((t & 3) == 3) {      // log of the product of four innovation variances at a time
            LS += log(prod);
            prod = 1.0;
        }
        bool moved = false, cyc = t >= 1;
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const double Pn = pp[i][j] - (V[i] * rs) * (V[j] * rs);
                moved = moved || fabs(Pn - P[i][j]) > kTol * 0.5 * (pp[i][i] + pp[j][j]);
                cyc = cyc && (Pn == Pold2[i][j]);
                Pold2[i][j] = P[i][j];
                P[i][j] = Pn;
            }
        if ((t & 63) == 63) gains(t - 63 + lane);
        if (!moved || cyc) tc = t;
    }
    LS += log(prod);
    if (n0 >= 0 && (n0 & ~63) + lane <= n0) gains((n0 & ~63) + lane);
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) Pss[i][j] = P[i][j];
    (void)ss;
    return n0;
}

// ---- phase (a) + (b), d <= 4, time-parallel: the filtered covariances of 64 consecutive steps at once --------------------------------
// The covariance recursion of an LTI model applies the SAME Riccati map at every step, and t-fold compositions of it are the
// covariance parts (A, C, J) of the filter scan's element E^t (Sarkka & Garcia-Fernandez 2021; the general engine's FilterMonoid):
//     P_t = A_t (I + P_in J_t)^-1 P_in A_t' + C_t.
// One wave forms E^1 .. E^64 by a Hillis-Steele scan over its lanes (six combines per lane instead of 63 dependent steps); lane l
// applies E^(l+1) to the covariance the block starts from and owns step 64 b + l: its predicted covariance, innovation variance, gains
// and reverse-time dynamics. The block that contains the first step whose filtered covariance has stopped moving to ~1e-14 hands over to
// the sequential recursion, which runs the last few steps to the exact fixed point (the criterion of filter_cov_reg, 2 ulp): the head is
// the same sequence of covariances to rounding, n0 the same kind of index, and only ~10 of its ~60 steps are sequential.
template <int D>
struct CovElem {
    double A[D][D], C[D][D], J[D][D];
};
template <int D>
__device__ __forceinline__ void ce_inverse(double (&M)[D][D], double (&X)[D][D]) {      // X = M^-1 (partial pivoting by selects); M destroyed
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) X[i][j] = (i == j) ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        int piv = k;
        double best = fabs(M[k][k]);
#pragma unroll
        for (int i = k + 1; i < D; ++i) {
            const double v = fabs(M[i][k]);
            const bool gt = v > best;
            best = gt ? v : best;
            piv = gt ? i : piv;
        }
#pragma unroll
        for (int i = k + 1; i < D; ++i) {
            const bool sw = piv == i;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const double t = M[k][j], u = M[i][j], t2 = X[k][j], u2 = X[i][j];
                M[k][j] = sw ? u : t;
                M[i][j] = sw ? t : u;
                X[k][j] = sw ? u2 : t2;
                X[i][j] = sw ? t2 : u2;
            }
        }
        const double inv = 1.0 / M[k][k];
#pragma unroll
        for (int i = k + 1; i < D; ++i) {
            const double f = M[i][k] * inv;
#pragma unroll
            for (int j = k + 1; j < D; ++j) M[i][j] = fma(-f, M[k][j], M[i][j]);
#pragma unroll
            for (int j = 0; j < D; ++j) X[i][j] = fma(-f, X[k][j], X[i][j]);
        }
    }
#pragma unroll
    for (int k = D - 1; k >= 0; --k) {
        const double inv = 1.0 / M[k][k];
#pragma unroll
        for (int j = 0; j < D; ++j) {
            double acc = X[k][j];
#pragma unroll
            for (int i = k + 1; i < D; ++i) acc = fma(-M[k][i], X[i][j], acc);
            X[k][j] = acc * inv;
        }
    }
}
template <int D, bool TX, bool TY>
__device__ __forceinline__ void ce_mul(const double (&X)[D][D], const double (&Y)[D][D], double (&Z)[D][D]) {      // Z = op(X) op(Y)
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) {
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) v = fma(TX ? X[k][i] : X[i][k], TY ? Y[j][k] : Y[k][j], v);
            Z[i][j] = v;
        }
}
template <int D>
__device__ __forceinline__ void ce_sym(double (&P)[D][D]) {
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = i + 1; j < D; ++j) {
            const double v = 0.5 * (P[i][j] + P[j][i]);
            P[i][j] = P[j][i] = v;
        }
}
// out = later o earlier
template <int D>
__device__ __forceinline__ void ce_combine(const CovElem<D>& ei, const CovElem<D>& ej, CovElem<D>& out) {
    double M[D][D], Mi[D][D], T1[D][D], T2[D][D], JA[D][D], MJA[D][D];
    ce_mul<D, false, false>(ei.C, ej.J, M);
#pragma unroll
    for (int i = 0; i < D; ++i) M[i][i] += 1.0;
    ce_inverse<D>(M, Mi);                                 // (I + C_i J_j)^-1
    ce_mul<D, false, false>(ej.A, Mi, T1);                // A_j M
    ce_mul<D, false, false>(ej.J, ei.A, JA);              // J_j A_i
    ce_mul<D, true, false>(Mi, JA, MJA);                  // M' J_j A_i
    double nA[D][D], nC[D][D], nJ[D][D];
    ce_mul<D, true, false>(ei.A, MJA, nJ);                // A_i' M' J_j A_i
    ce_mul<D, false, false>(T1, ei.A, nA);
    ce_mul<D, false, false>(T1, ei.C, T2);
    ce_mul<D, false, true>(T2, ej.A, nC);
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) {
            out.A[i][j] = nA[i][j];
            out.C[i][j] = nC[i][j] + ej.C[i][j];
            out.J[i][j] = nJ[i][j] + ei.J[i][j];
        }
    ce_sym<D>(out.C);
    ce_sym<D>(out.J);
}
template <int D>
__device__ __forceinline__ void ce_apply(const CovElem<D>& e, const double (&P)[D][D], double (&out)[D][D]) {
    double M[D][D], Mi[D][D], T1[D][D], T2[D][D];
    ce_mul<D, false, false>(P, e.J, M);
#pragma unroll
    for (int i = 0; i < D; ++i) M[i][i] += 1.0;
    ce_inverse<D>(M, Mi);
    ce_mul<D, false, false>(e.A, Mi, T1);
    ce_mul<D, false, false>(T1, P, T2);
    ce_mul<D, false, true>(T2, e.A, out);
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) out[i][j] += e.C[i][j];
    ce_sym<D>(out);
}

template <int D>
__device__ __forceinline__ int filter_cov_scan(const ModelDev& m, const Tab& tb, int lane, bool& bad, double (&Pss)[D][D], double (&kAss)[D],
                                               double& Sss, double& LS) {
    constexpr int DD = D * D;
    constexpr int nhmax = kHeadMaxTiles * kTile;
    constexpr double kTolScan = 256.0 * kTol, kTolFine = 16.0 * kTol;
    double A[D][D], Q[D][D], hv[D], P[D][D], Pold2[D][D], cPf[D][D], cPp[D][D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        hv[i] = m.H[i];
#pragma unroll
        for (int k = 0; k < D; ++k) {
            A[i][k] = m.A[i + k * D];
            const int r = i < k ? i : k, c = i < k ? k : i;
            Q[i][k] = m.Q[r + c * D];                        // Symmetric(Q), Symmetric(x0P): upper triangles
            P[i][k] = m.x0[D + c * (c + 1) / 2 + r];
            Pold2[i][k] = 0.0;
            cPf[i][k] = cPp[i][k] = 0.0;
        }
    }
    const double R = m.R[0];
    auto gains = [&](int t) {        // reverse-time dynamics and smoother gain of the step this lane has kept
        double G[D][D], L[D][D];
        const bool ok = invert_dynamics_lane<D>(m.A, cPf, cPp, G, L);
        if (!ok) tb.hdr[4] = 1;
        double K[D], Sv = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            double v = 0.0;
#pragma unroll
            for (int l = 0; l < D; ++l) v = fma(hv[l], cPp[l][k], v);
            K[k] = v;
            Sv = fma(v, hv[k], Sv);
        }
        Sv += R;
        const double iSv = 1.0 / Sv;
#pragma unroll
        for (int r = 0; r < D; ++r) {
            double v = 0.0;
#pragma unroll
            for (int c = 0; c < D; ++c) {
                v = fma(G[r][c], K[c] * iSv, v);
                if (t < nhmax) tb.h_G[(size_t)t * DD + r * D + c] = G[r][c];
                tb.s_L[(size_t)t * DD + r * D + c] = L[r][c];
            }
            if (t < nhmax) tb.h_c[t * D + r] = v;
        }
    };
    // one step from the filtered covariance Pin: predicted covariance, V = Pp h, S
    auto one_step = [&](const double (&Pin)[D][D], double (&pp)[D][D], double (&V)[D], double& S) {
        double t1[D][D];
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) {
                double v = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) v = fma(A[i][k], (k <= j ? Pin[k][j] : Pin[j][k]), v);       // A * Symmetric(P)
                t1[i][j] = v;
            }
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) {
                double v = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) v = fma(t1[i][k], A[j][k], v);
                pp[i][j] = v + Q[i][j];
            }
        S = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            double v = 0.0;
#pragma unroll
            for (int l = 0; l < D; ++l) v = fma(hv[l], pp[l][k], v);
            V[k] = v;
            S = fma(v, hv[k], S);
        }
        S += R;
    };
    // ---- the one-step element and its powers E^(lane + 1)
    CovElem<D> pre;
    {
        double Qh[D], S1 = R;
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) v = fma(Q[i][k], hv[k], v);
            Qh[i] = v;
        }
#pragma unroll
        for (int i = 0; i < D; ++i) S1 = fma(hv[i], Qh[i], S1);
        bad = bad || !(S1 > 0.0);
        const double iS1 = 1.0 / S1;
        double Ath[D];                       // A' h
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) v = fma(A[k][i], hv[k], v);
            Ath[i] = v;
        }
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) {
                // (I - K h') X = X - K (h' X),  K = Q h / S1
                pre.A[i][j] = fma(-Qh[i] * iS1, Ath[j], A[i][j]);
                pre.C[i][j] = fma(-Qh[i] * iS1, Qh[j], Q[i][j]);
                pre.J[i][j] = Ath[i] * Ath[j] * iS1;
            }
    }
#pragma unroll 1
    for (int k = 0; k < 6; ++k) {
        const int off = 1 << k;
        CovElem<D> other, res;
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) {
                other.A[i][j] = __shfl_up(pre.A[i][j], off);
                other.C[i][j] = __shfl_up(pre.C[i][j], off);
                other.J[i][j] = __shfl_up(pre.J[i][j], off);
            }
        ce_combine<D>(other, pre, res);      // (every lane computes; lanes below `off` keep their own)
        if (lane >= off) pre = res;
    }
    // ---- blocks of 64 steps until the covariance has (nearly) stopped moving
    int tstar = -1;
    LS = 0.0;
#pragma unroll 1
    for (int b = 0; b * 64 < nhmax; ++b) {
        const int t = b * 64 + lane;
        double Pf[D][D], Pprev[D][D], pp[D][D], V[D], S;
        ce_apply<D>(pre, P, Pf);                                   // filtered covariance behind step t
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const double up = __shfl_up(Pf[i][j], 1);
                Pprev[i][j] = lane == 0 ? P[i][j] : up;             // ... and in front of it
            }
        one_step(Pprev, pp, V, S);
        const bool okS = S > 0.0;
        bool moved = false, moved_fine = false;
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const double dlt = fabs(Pf[i][j] - Pprev[i][j]), sc = 0.5 * (pp[i][i] + pp[j][j]);
                moved = moved || !(dlt <= kTolScan * sc);
                moved_fine = moved_fine || !(dlt <= kTolFine * sc);
                cPf[i][j] = Pprev[i][j];
                cPp[i][j] = pp[i][j];
            }
        bad = bad || __any(!okS);
        // the first step of the block that no longer moves at the fine tolerance (few sequential steps are left from there); if the
        // block's values only agree to the coarse one -- rounding of the composed elements -- its last step (the most converged one)
        const unsigned long long still_fine = __ballot(!moved_fine), still = __ballot(!moved);
        const int lstar = still_fine ? __ffsll((long long)still_fine) - 1 : (still ? 63 : 64);
        const double iS = 1.0 / S;
        if (lane <= lstar && lane < 64) {
#pragma unroll
            for (int i = 0; i < D; ++i) {
                double v = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) v = fma(A[i][k], V[k] * iS, v);
                tb.h_kA[t * D + i] = v;
            }
            tb.h_rS[t] = R * iS;
            tb.h_iS[t] = iS;
        }
        double ls = (lane <= lstar) ? log(S) : 0.0;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) ls += __shfl_xor(ls, off);
        LS += ls;
        if (lstar < 64) {
            tstar = b * 64 + lstar;
            if (lstar == 63) gains(t);                              // (the block is complete: the sequential steps start in the next one)
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) P[i][j] = __shfl(Pf[i][j], lstar);
            break;
        }
        gains(t);                                                   // the block is complete
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) P[i][j] = __shfl(Pf[i][j], 63);
    }
    if (tstar < 0 || bad) return -1;
    // ---- the last steps to the exact fixed point, sequentially (every lane the same arithmetic; lane t & 63 keeps step t)
    int tc = -1, n0 = -1, done = 0;
    double prod = 1.0;
#pragma unroll 1
    for (int t = tstar + 1; t <= nhmax; ++t) {
        double pp[D][D], V[D], S;
        one_step(P, pp, V, S);
        bad = bad || !(S > 0.0);
        const double iS = 1.0 / S, rs = 1.0 / sqrt(S);
        const bool mine = (t & 63) == lane;
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) v = fma(A[i][k], V[k] * iS, v);
            kAss[i] = v;
            if (mine && t < nhmax) tb.h_kA[t * D + i] = v;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                cPf[i][j] = mine ? P[i][j] : cPf[i][j];
                cPp[i][j] = mine ? pp[i][j] : cPp[i][j];
            }
        }
        if (mine && t < nhmax) {
            tb.h_rS[t] = R * iS;
            tb.h_iS[t] = iS;
        }
        Sss = S;
        if (tc >= 0) {           // the extra iteration from the settled covariance: the stationary step
            n0 = t;
            break;
        }
        if (t == nhmax) break;
        prod *= S;
        if ((done & 3) == 3) {
            LS += log(prod);
            prod = 1.0;
        }
        bool moved = false, cyc = done >= 1;
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const double Pn = pp[i][j] - (V[i] * rs) * (V[j] * rs);
                moved = moved || fabs(Pn - P[i][j]) > kTol * 0.5 * (pp[i][i] + pp[j][j]);
                cyc = cyc && (Pn == Pold2[i][j]);
                Pold2[i][j] = P[i][j];
                P[i][j] = Pn;
            }
        if ((t & 63) == 63) gains(t - 63 + lane);
        if (!moved || cyc) tc = t;
        ++done;
    }
    LS += log(prod);
    if (n0 >= 0 && (n0 & ~63) + lane <= n0) gains((n0 & ~63) + lane);
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) Pss[i][j] = P[i][j];
    return n0;
}

// ---- phase (a) + (b), d >= 4: lane (i, j) owns element (i, j) of every matrix, LDS is the exchange (a d x d product costs a lane d
// multiply-adds instead of d^3); the steps' (Pf, Pp) go to scratch and the gains are computed afterwards, one step per lane.
template <int D>
__device__ __forceinline__ int filter_cov_lds(const ModelDev& m, const Tab& tb, int lane, bool& bad, double* sP, double* sT, double* sPp,
                                              double (&kAss)[D], double& Sss, double& LS) {
    constexpr int DD = D * D;
    constexpr int nhmax = kHeadMaxTiles * kTile;
    const bool act = lane < DD;
    const int e = act ? lane : 0;
    const int i = e / D, j = e % D;
    __shared__ double sKA[D], sV[D];
    double Ai[D], Aj[D], hv[D];
#pragma unroll
    for (int k = 0; k < D; ++k) {
        Ai[k] = m.A[i + k * D];
        Aj[k] = m.A[j + k * D];
        hv[k] = m.H[k];
    }
    const double Qij = m.Q[i + j * D];
    const double R = m.R[0];
    {
        const int r = imin(i, j), c = imax(i, j);
        if (act) sP[e] = m.x0[D + c * (c + 1) / 2 + r];
    }
    lds_sync();
    double Pold2 = 0.0, prod = 1.0;
    int tc = -1, n0 = -1;
    LS = 0.0;
    for (int t = 0; t <= nhmax; ++t) {
        const double Pf = sP[e];
        double t1 = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) t1 = fma(Ai[k], sP[imin(k, j) * D + imax(k, j)], t1);     // A * Symmetric(P): upper triangle
        if (act) sT[e] = t1;
        lds_sync();
        double pp = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) pp = fma(sT[i * D + k], Aj[k], pp);
        pp += Qij;
        if (act) sPp[e] = pp;
        lds_sync();
        double V[D], S = 0.0;
        {
            double v = 0.0;                                                      // V = H * Pp: lane (0, k) forms V_k, everybody reads the d values
#pragma unroll
            for (int l = 0; l < D; ++l) v = fma(hv[l], sPp[l * D + j], v);
            if (act && i == 0) sV[j] = v;
        }
        lds_sync();
#pragma unroll
        for (int k = 0; k < D; ++k) {
            V[k] = sV[k];
            S = fma(V[k], hv[k], S);
        }
        S += R;
        bad = bad || !(S > 0.0);
        const double iS = 1.0 / S, rs = 1.0 / sqrt(S);
        const double Pn = pp - (V[i] * rs) * (V[j] * rs);
        // the step's gain row by row: lane (i, 0) owns (A K)_i (its own row of A is in registers); the lanes pick up the whole
        // stationary vector once, behind the loop (a d x d product and d^2 loads of A per step in EVERY lane cost a third of the step)
        double kAi = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) kAi = fma(Ai[k], V[k] * iS, kAi);
        if (act && j == 0) sKA[i] = kAi;
        if (t < nhmax) {
            if (act && j == 0) tb.h_kA[t * D + i] = kAi;
            if (lane == 0) {
                tb.h_rS[t] = R * iS;
                tb.h_iS[t] = iS;
            }
        }
        if (act) {
            tb.s_Pf[(size_t)t * DD + e] = Pf;
            tb.s_Pp[(size_t)t * DD + e] = pp;
        }
        Sss = S;
        if (tc >= 0) {
            n0 = t;
            break;
        }
        if (t == nhmax) break;
        prod *= S;
        if ((t & 3) == 3) {
            LS += log(prod);
            prod = 1.0;
        }
        const double scale = 0.5 * (sPp[i * D + i] + sPp[j * D + j]);
        const bool moved = act && fabs(Pn - Pf) > kTol * scale;
        const bool nocyc = act && !(Pn == Pold2);
        const bool conv = !__any(moved) || (t >= 1 && !__any(nocyc));
        Pold2 = Pf;
        lds_sync();
        if (act) sP[e] = Pn;
        lds_sync();
        if (conv) tc = t;
    }
    LS += log(prod);
    lds_sync();
#pragma unroll
    for (int r = 0; r < D; ++r) kAss[r] = sKA[r];            // (the last step's: the stationary one when the loop ended by convergence)
    if (n0 < 0) return n0;
    __threadfence_block();
    __syncthreads();
    for (int t0 = 0; t0 <= n0; t0 += 64) {
        const int t = t0 + lane;
        if (t <= n0) {
            double Pf[D][D], Pp[D][D], G[D][D], L[D][D];
#pragma unroll
            for (int r = 0; r < D; ++r)
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    Pf[r][c] = tb.s_Pf[(size_t)t * DD + r * D + c];
                    Pp[r][c] = tb.s_Pp[(size_t)t * DD + r * D + c];
                }
            const bool ok = invert_dynamics_lane<D>(m.A, Pf, Pp, G, L);
            if (!ok) tb.hdr[4] = 1;
            double K[D], Sv = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) {
                double v = 0.0;
#pragma unroll
                for (int l = 0; l < D; ++l) v = fma(hv[l], Pp[l][k], v);
                K[k] = v;
                Sv = fma(v, hv[k], Sv);
            }
            Sv += R;
            const double iSv = 1.0 / Sv;
#pragma unroll
            for (int r = 0; r < D; ++r) {
                double v = 0.0;
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    v = fma(G[r][c], K[c] * iSv, v);
                    if (t < nhmax) tb.h_G[(size_t)t * DD + r * D + c] = G[r][c];
                    tb.s_L[(size_t)t * DD + r * D + c] = L[r][c];
                }
                if (t < nhmax) tb.h_c[t * D + r] = v;
            }
        }
    }
    return n0;
}

template <int D>
__device__ void head_forward(const Tab& tb, const double* __restrict__ y, long long T, int lane);

// k_setup_core: what the tile passes wait for -- the filter covariance to its stationary value with the head's per-step gains (a, b),
// the constant block with the powers and couplings (e), and the head's forward recursion (its carry starts the stationary tiles).
template <int D>
__device__ void setup_side_scan(const ModelDev& m, const Tab& tb, long long T);
// (d <= kCovScanMaxD: launched with TWO waves. Wave 1 waits at the one workgroup barrier for the head's covariances and gains and then
//  builds the variance tables of a posterior call -- setup_side_scan -- beside wave 0's later phases; everything else in here is wave 0
//  talking to itself, hence wave_sync, not __syncthreads.)
template <int D>
__device__ __forceinline__ void setup_body(const ModelDev& m, const Tab& tb, const double* __restrict__ y, long long T, int grad, int shard, int with_side,
                                           double* __restrict__ result) {
    constexpr int DD = D * D;
    constexpr int nhmax = kHeadMaxTiles * kTile;
    __shared__ double sP[DD], sT[DD], sPp[DD], sB[DD], sXa[DD], sGa[DD], sXb[DD], sGb[DD], sXc[DD], sGc[DD];
    if (threadIdx.x >= 64) {
        __syncthreads();
        if constexpr (D <= kCovScanMaxD) {
            if (with_side && tb.hdr[0] != 0 && (tb.hdr[6] & 4) == 0) setup_side_scan<D>(m, tb, T);
        }
        return;
    }
    const int lane = threadIdx.x;
    const bool act = lane < DD;
    const int e = act ? lane : 0;
    const int i = e / D, j = e % D;
    if (result != nullptr && lane < 8) result[lane] = 0.0;      // the call's result record (the API layer then skips its own clearing launch)
    if (lane == 0) {
        tb.hdr[4] = 0;
        tb.misc[8] = (double)wall_clock64();
    }
    wave_sync();
    bool bad = false;
    double kAss[D], Sss = 1.0, LS = 0.0;
    int n0;
    const bool scan_cov = (shard & 4) == 0;      // (bit 2 of the mode word: the sequential covariance iteration, for A/B runs)
    if (D <= kCovScanMaxD && (scan_cov || D <= 3)) {
        double Pss[D][D];
        if constexpr (D <= kCovScanMaxD) {
            if (scan_cov) n0 = filter_cov_scan<D>(m, tb, lane, bad, Pss, kAss, Sss, LS);
            else if constexpr (D <= 3) n0 = filter_cov_reg<D>(m, tb, lane, bad, Pss, kAss, Sss, LS);
            else n0 = -1;
        } else {
            n0 = -1;
        }
        if (act) {      // the stationary filtered covariance, for k_setup_side
            double v = 0.0;
#pragma unroll
            for (int r = 0; r < D; ++r)
#pragma unroll
                for (int c = 0; c < D; ++c) v = (r == i && c == j) ? Pss[r][c] : v;
            tb.s_Pf[(size_t)(n0 >= 0 ? n0 : 0) * DD + e] = v;
        }
    } else {
        n0 = filter_cov_lds<D>(m, tb, lane, bad, sP, sT, sPp, kAss, Sss, LS);
    }
    __threadfence_block();
    wave_sync();
    bad = bad || tb.hdr[4] != 0;
    const bool settled = n0 >= 0 && n0 < nhmax;      // the head tables hold nhmax steps: th <= kHeadMaxTiles
    // A time shard that does not start the series (shard & 1) has no head: every one of its steps is stationary, its first predicted
    // mean comes from the exchange. One that does not end it (shard & 2) hands its end state on: that needs whole tiles, and the whole
    // run of stationary steps as ONE element (Phi^L, G^L, B_L below: L < 2^kPowN).
    const bool notfirst = (shard & 1) != 0, notlast = (shard & 2) != 0;
    shard &= 3;
    const int th = settled ? (notfirst ? 0 : n0 / kTile + 1) : 0;
    const int nh = th * kTile;
    const long long Lseg = T - nh;
    const bool fits = (long long)nh + 2 <= T && (!shard || Lseg < (1LL << kPowN)) && (!notlast || T % kTile == 0);
    if (lane == 0) {
        tb.hdr[0] = (settled && !bad && fits) ? 1 : 0;      // (k_setup_side may still find the series too short)
        tb.hdr[1] = th;
        tb.hdr[2] = settled ? (notfirst ? 0 : n0) : -1;      // head steps with gains of their own (what the passes and the caller see)
        tb.hdr[3] = -1;
        tb.hdr[4] = bad ? 1 : 0;
        tb.hdr[5] = n0;                                     // row of the tables that holds the stationary step
        tb.hdr[6] = shard | (scan_cov ? 0 : 4);          // bits 0, 1: the shard's missing ends; bit 2: sequential covariance recursions (A/B)
        tb.misc[9] = (double)wall_clock64();
    }
    __threadfence_block();
    __syncthreads();          // the head's covariances and gains are in memory: wave 1 (if the launch has one) starts on the variance tables
    if (!settled || bad || !fits) return;
    // stationary coefficients (entry n0; the head recursions read the tables at min(t, n0))
    double* ss = tb.ssc;
    const double iS = 1.0 / Sss, R = m.R[0];
    double hv[D], Ai[D];
#pragma unroll
    for (int k = 0; k < D; ++k) {
        hv[k] = m.H[k];
        Ai[k] = m.A[i + k * D];
    }
    // Adjoint (gradient) calls run the SAME backward machinery on the adjoint of the predicted mean instead of the smoother's lam:
    //   psi_t = Phi' psi_{t+1} + h r_t / S   (psi_t = d logpdf / d mu_t at fixed gains)   <->   lam <- G lam + c r
    // so the block's (G, c) become (Phi', h / S) and every power / coupling below follows from them unchanged.
    if (act) {
        ss[SS<D>::A + e] = Ai[j];
        ss[SS<D>::G + e] = grad ? (m.A[j + i * D] - kAss[j] * hv[i]) : tb.h_G[(size_t)n0 * DD + e];
    }
    if (act && j == 0) {
        ss[SS<D>::a + i] = m.a[i];
        ss[SS<D>::h + i] = hv[i];
        double v = m.a[i];
#pragma unroll
        for (int k = 0; k < D; ++k) v = fma(Ai[k], m.x0[k], v);
        ss[SS<D>::mu0 + i] = v;
        ss[SS<D>::c + i] = grad ? hv[i] * iS : tb.h_c[n0 * D + i];
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < D; ++k) ss[SS<D>::kA + k] = kAss[k];
        ss[SS<D>::hh] = m.hh[0];
        ss[SS<D>::rS] = R * iS;
        ss[SS<D>::iS] = iS;
        ss[SS<D>::logS] = log(Sss);
        ss[SS<D>::LS] = notfirst ? 0.0 : LS;
    }
    __threadfence_block();
    wave_sync();
    if (lane == 0) tb.misc[10] = (double)wall_clock64();
    // ---- rows h' Phi^j, j < 16 (tile_forward / half_tile_forward: the lanes' innovations from their scanned start state without a second recursion)
    if (lane < D) {
        double w[D], nw[D];
#pragma unroll
        for (int k = 0; k < D; ++k) w[k] = hv[k];
#pragma unroll
        for (int jj = 0; jj < 2 * kSub; ++jj) {
#pragma unroll
            for (int c = 0; c < D; ++c) {
                if (c == lane) tb.cst[CL<D>::WJ + jj * D + c] = w[c];
                double v = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) v = fma(w[k], m.A[k + c * D] - kAss[k] * hv[c], v);      // (w' Phi)_c
                nw[c] = v;
            }
#pragma unroll
            for (int c = 0; c < D; ++c) w[c] = nw[c];
        }
    }
    // ---- (e) powers Phi^(2^k), G^(2^k); the couplings B_n = sum_{j < n} G^j c h' Phi^j by doubling, B_2n = B_n + G^n B_n Phi^n:
    //      B_512 (a tile), B_2048 (a workgroup), and the two ragged ones -- the last tile's nv valid steps and the last workgroup's nvb --
    //      composed from the B_(2^k) of the bits of nv / nvb --------------------------------------------------------------------------------
    {
        const long long ntiles = (T + kTile - 1) / kTile;
        const long long nblk = (ntiles - th + kBlk - 1) / kBlk;
        const int nv = (int)(T - (ntiles - 1) * kTile);                                  // 1..512
        const int nvb = (int)(T - ((long long)th + (nblk - 1) * kBlk) * kTile);          // 1..2048
        const double ci = ss[SS<D>::c + i];
        double x = Ai[j] - kAss[i] * hv[j];        // Phi = A - kA h'
        double g = ss[SS<D>::G + e];
        double b = ci * hv[j];
        double bl[3] = {0.0, 0.0, 0.0};
        const long long want[3] = {nv, nvb, shard ? Lseg : 0LL};
        double* sXq[3] = {sXa, sXb, sXc};
        double* sGq[3] = {sGa, sGb, sGc};
        if (act) {
            sXa[e] = sGa[e] = sXb[e] = sGb[e] = sXc[e] = sGc[e] = (i == j) ? 1.0 : 0.0;
        }
        lds_sync();
        double* cst = tb.cst;
        for (int k = 0; k < kPowN; ++k) {
            if (k == kLogBlk + 1 && !shard) {
                // short memory (hdr[7] below): nothing behind a whole workgroup's power is ever read -- k_carry needs no scan
                const double px = act ? fabs(cst[CL<D>::pphi + kLogBlk * DD + e]) : 0.0, gx = act ? fabs(cst[CL<D>::pg + kLogBlk * DD + e]) : 0.0;
                if (!__any((px > 1e-30) || (gx > 1e-30))) break;
            }
            if (act) {
                cst[CL<D>::pphi + k * DD + e] = x;
                cst[CL<D>::pg + k * DD + e] = g;
                if (k == kLogTile) cst[CL<D>::B512 + e] = b;
                if (k == kLogBlk) cst[CL<D>::Bblk + e] = b;
                sP[e] = x;
                sT[e] = g;
                sB[e] = b;
            }
            lds_sync();
            if (k < kLogBlk || shard) {             // (the tile / workgroup targets have no bits from 2^kLogBlk on)
#pragma unroll
                for (int w = 0; w < 3; ++w) {
                    if ((want[w] >> k) & 1) {       // append a block of 2^k steps to the composition: Bl += Ga B_(2^k) Xa
                        double u = 0.0, xa = 0.0, ga = 0.0;
#pragma unroll
                        for (int l = 0; l < D; ++l) {
                            u = fma(sB[i * D + l], sXq[w][l * D + j], u);
                            xa = fma(sP[i * D + l], sXq[w][l * D + j], xa);
                            ga = fma(sT[i * D + l], sGq[w][l * D + j], ga);
                        }
                        if (act) sPp[e] = u;
                        lds_sync();
                        double v = 0.0;
#pragma unroll
                        for (int l = 0; l < D; ++l) v = fma(sGq[w][i * D + l], sPp[l * D + j], v);
                        bl[w] += v;
                        lds_sync();
                        if (act) {
                            sXq[w][e] = xa;
                            sGq[w][e] = ga;
                        }
                        lds_sync();
                    }
                }
                double u = 0.0;
#pragma unroll
                for (int l = 0; l < D; ++l) u = fma(sB[i * D + l], sP[l * D + j], u);      // B Phi^m
                if (act) sPp[e] = u;
                lds_sync();
                double v = 0.0;
#pragma unroll
                for (int l = 0; l < D; ++l) v = fma(sT[i * D + l], sPp[l * D + j], v);     // G^m (B Phi^m)
                b += v;
            }
            double x2 = 0.0, g2 = 0.0;
#pragma unroll
            for (int l = 0; l < D; ++l) {
                x2 = fma(sP[i * D + l], sP[l * D + j], x2);
                g2 = fma(sT[i * D + l], sT[l * D + j], g2);
            }
            x = x2;
            g = g2;
            lds_sync();
        }
        __threadfence_block();
        wave_sync();
        if (act) {
            cst[CL<D>::Blt + e] = (nv == kTile) ? cst[CL<D>::B512 + e] : bl[0];
            cst[CL<D>::Blb + e] = (nvb == kBlk * kTile) ? cst[CL<D>::Bblk + e] : bl[1];
            cst[CL<D>::PSeg + e] = sXc[e];
            cst[CL<D>::GSeg + e] = sGc[e];
            cst[CL<D>::BSeg + e] = bl[2];
            cst[CL<D>::GLb + e] = (nvb == kBlk * kTile) ? cst[CL<D>::pg + kLogBlk * DD + e] : sGb[e];
        }
        {
            // Short memory: when Phi^4096 and G^4096 (a whole workgroup's steps) have decayed to nothing, a workgroup's carries depend on
            // its neighbours' elements only and k_carry needs no scan (hdr[7]; every well-mixing model: 0.99^4096 = 1e-18, 0.98^4096 = 1e-36).
            const double px = act ? fabs(cst[CL<D>::pphi + kLogBlk * DD + e]) : 0.0, gx = act ? fabs(cst[CL<D>::pg + kLogBlk * DD + e]) : 0.0;
            const bool tiny = !(px > 1e-30) && !(gx > 1e-30);
            const bool all_tiny = !__any(!tiny);
            if (lane == 0) tb.hdr[7] = all_tiny ? 1 : 0;
        }
        // (e2) tile carries inside a workgroup in closed form (k_apply): PT[w] = Phi^(512 w), GT[w] = G^(512 w), and the coupling of the
        //      workgroup's mu into the lam behind tile w, K_w = G^512 K_{w+1} + B_512 PT[w+1], K_{kBlk-1} = 0
        {
            const double x5 = cst[CL<D>::pphi + kLogTile * DD + e], g5 = cst[CL<D>::pg + kLogTile * DD + e], b5 = cst[CL<D>::B512 + e];
            if (act) {
                sP[e] = x5;
                sT[e] = g5;
                sB[e] = b5;
                sXa[e] = sGa[e] = (i == j) ? 1.0 : 0.0;      // running PT, GT
                sXb[e] = 0.0;                                // running K
            }
            lds_sync();
            double ptw[kBlk];                                // this lane's element of PT[w]
            for (int w = 0; w < kBlk; ++w) {
                const double pt = sXa[e], gt = sGa[e];
                ptw[w] = pt;
                if (act) {
                    cst[CL<D>::PT + w * DD + e] = pt;
                    cst[CL<D>::GT + w * DD + e] = gt;
                }
                double pn = 0.0, gn = 0.0;
#pragma unroll
                for (int l = 0; l < D; ++l) {
                    pn = fma(sXa[i * D + l], sP[l * D + j], pn);
                    gn = fma(sGa[i * D + l], sT[l * D + j], gn);
                }
                lds_sync();
                if (act) {
                    sXa[e] = pn;
                    sGa[e] = gn;
                }
                lds_sync();
            }
            if (act) cst[CL<D>::KW + (kBlk - 1) * DD + e] = 0.0;
            for (int w = kBlk - 2; w >= 0; --w) {
                if (act) sXa[e] = ptw[w + 1];
                lds_sync();
                double kn = 0.0;
#pragma unroll
                for (int l = 0; l < D; ++l) {
                    kn = fma(sT[i * D + l], sXb[l * D + j], kn);        // G^512 K_{w+1}
                    kn = fma(sB[i * D + l], sXa[l * D + j], kn);        // + B_512 PT[w+1]
                }
                lds_sync();
                if (act) {
                    sXb[e] = kn;
                    cst[CL<D>::KW + w * DD + e] = kn;
                }
                lds_sync();
            }
        }
    }
    __threadfence_block();
    wave_sync();
    if (lane == 0) tb.misc[11] = (double)wall_clock64();
    if (notfirst) {      // no head: a zero carry-in until the exchange supplies the real one (k_shard_fold)
        if (lane == 0) {
            tb.misc[0] = 0.0;
#pragma unroll
            for (int q = 0; q < D; ++q) tb.MUb[q] = 0.0;
        }
    } else {
        head_forward<D>(tb, y, T, lane);
    }
    if (lane == 0) tb.misc[12] = (double)wall_clock64();
}

template <int D>
__global__ __launch_bounds__(128) void k_setup_core(ModelDev m, Tab tb, const double* __restrict__ y, long long T, int grad, int shard, int with_side,
                                                    double* __restrict__ result) {
    setup_body<D>(m, tb, y, T, grad, shard, with_side, result);
}

// setup_side: the smoothed VARIANCES of the head and of the tail (c, d) -- needed by the output pass only, so ONE wave of an extra
// workgroup of pass 1 computes them beside the tiles (lane per matrix element: few registers, it does not lower pass 1's occupancy).
template <int D>
__device__ void setup_side(const ModelDev& m, const Tab& tb, long long T) {
    constexpr int DD = D * D;
    constexpr int SB = D <= 6 ? 64 : 32;          // steps of the head staged in LDS at a time (phase d)
    __shared__ double sP[DD], sT[DD];
    __shared__ double sGt[SB * DD], sLt[SB * DD];
    const int lane = threadIdx.x;
    const bool act = lane < DD;
    const int e = act ? lane : 0;
    const int i = e / D, j = e % D;
    const int th = (int)tb.hdr[1], n0 = (int)tb.hdr[5];
    const bool notfirst = (tb.hdr[6] & 1) != 0, notlast = (tb.hdr[6] & 2) != 0;      // time shards: no head / no tail of its own
    const int nh = th * kTile;
    const double* ss = tb.ssc;
    double hv[D];
#pragma unroll
    for (int k = 0; k < D; ++k) hv[k] = m.H[k];
    if (lane == 0) tb.misc[13] = (double)wall_clock64();
    if (act) sP[e] = tb.s_Pf[(size_t)n0 * DD + e];         // P_ss
    lds_sync();
    double Pold2 = 0.0;
    const bool bad = false;
    // ---- (c) smoothed covariance backwards from the final filtered state (Ps_{T-1} = P_ss) until it no longer changes --------------
    double Gi[D], Gj[D];
#pragma unroll
    for (int k = 0; k < D; ++k) {
        Gi[k] = ss[SS<D>::G + i * D + k];
        Gj[k] = ss[SS<D>::G + j * D + k];
    }
    const double Lij = tb.s_L[(size_t)n0 * DD + e];
    int n1 = -1;
    for (int jt = 0; jt < kTailMax; ++jt) {
        const double Ps = sP[e];
        if (act) tb.t_Ps[(size_t)jt * DD + e] = Ps;
        double t1 = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) t1 = fma(Gi[k], sP[imin(k, j) * D + imax(k, j)], t1);
        const double dii = fabs(sP[i * D + i]), djj = fabs(sP[j * D + j]);
        if (act) sT[e] = t1;
        lds_sync();
        double pn = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) pn = fma(sT[i * D + k], Gj[k], pn);
        pn += Lij;
        const bool moved = act && fabs(pn - Ps) > kTol * 0.5 * (dii + djj);
        const bool nocyc = act && !(pn == Pold2);
        const bool conv = !__any(moved) || (jt >= 1 && !__any(nocyc));
        Pold2 = Ps;
        lds_sync();
        if (act) sP[e] = pn;
        lds_sync();
        if (conv) {
            n1 = jt + 1;
            break;
        }
    }
    const bool applies = !bad && n1 >= 0 && (long long)nh + (notlast ? 0 : n1) + 1 <= T;
    if (lane == 0) {
        tb.hdr[3] = notlast ? 0 : n1;      // (a segment that does not end the series: the smoothed covariance is stationary to its last step)
        if (!applies) tb.hdr[0] = 0;
    }
    if (!applies) return;

    // ---- (d) smoothed covariance of the head: Ps_{n0} = stationary, Ps_{t-1} = G_t Ps_t G_t' + L_t; the gains of SB steps at a time
    //      are staged in LDS (a dependent global load per step would cost more than the step) --------------------------------------
    if (act) tb.s_Ps[(size_t)n0 * DD + e] = sP[e];
    for (int thi = notfirst ? 0 : n0; thi >= 1; thi -= SB) {
        const int tlo = imax(thi - SB + 1, 1);          // steps tlo..thi, thi first
        const int cnt = thi - tlo + 1;
        for (int idx = lane; idx < cnt * DD; idx += 64) {
            sGt[idx] = tb.h_G[(size_t)tlo * DD + idx];
            sLt[idx] = tb.s_L[(size_t)tlo * DD + idx];
        }
        lds_sync();
        for (int t = thi; t >= tlo; --t) {
            const double* gt = &sGt[(t - tlo) * DD];
            double t1 = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) t1 = fma(gt[i * D + k], sP[imin(k, j) * D + imax(k, j)], t1);
            if (act) sT[e] = t1;
            lds_sync();
            double pn = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) pn = fma(sT[i * D + k], gt[j * D + k], pn);
            pn += sLt[(t - tlo) * DD + e];
            lds_sync();
            if (act) {
                sP[e] = pn;
                tb.s_Ps[(size_t)(t - 1) * DD + e] = pn;
            }
            lds_sync();
        }
    }
    __threadfence_block();
    lds_sync();

    if (lane == 0) tb.misc[14] = (double)wall_clock64();
    // ---- (d2) variances, one step per lane; head entries beyond n0 repeat the stationary one -----------------------------------------
    auto quad = [&](const double* Pm) {     // H Symmetric(P) H'
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < D; ++c) {
            double v = 0.0;
#pragma unroll
            for (int r = 0; r < D; ++r) v = fma(hv[r], Pm[imin(r, c) * D + imax(r, c)], v);
            s = fma(v, hv[c], s);
        }
        return s;
    };
    const double vb_ss = quad(&tb.s_Ps[(size_t)n0 * DD]);
    if (lane == 0) tb.ssc[SS<D>::vb] = vb_ss;
    for (int t = lane; t < n1; t += 64) tb.t_vb[t] = quad(&tb.t_Ps[(size_t)t * DD]);
    for (int t = lane; t <= n0; t += 64) tb.h_vb[t] = quad(&tb.s_Ps[(size_t)t * DD]);
    if (lane == 0) tb.misc[15] = (double)wall_clock64();
}

// setup_side for d <= kCovScanMaxD, time-parallel: both covariance recursions of the smoother are LINEAR in the covariance,
//     Ps_{t-1} = G_t Ps_t G_t' + L_t,
// so a run of steps composes to one (M, S) pair -- Ps_out = M Ps_in M' + S, later o earlier = (M_l M_e, M_l S_e M_l' + S_l) -- and a
// wave forms the 64 prefix compositions of a block by a Hillis-Steele scan over its lanes. The tail uses the stationary (G, L) in every
// lane, the head each step's own (G_t, L_t) from the tables; lane l then evaluates its own step's variance H Ps H'.
template <int D>
struct AffCov {
    double M[D][D], S[D][D];
};
template <int D>
__device__ __forceinline__ void ac_combine(const AffCov<D>& e, const AffCov<D>& l, AffCov<D>& out) {      // out = l o e
    double T1[D][D], T2[D][D];
    ce_mul<D, false, false>(l.M, e.M, out.M);
    ce_mul<D, false, false>(l.M, e.S, T1);
    ce_mul<D, false, true>(T1, l.M, T2);
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) out.S[i][j] = T2[i][j] + l.S[i][j];
    ce_sym<D>(out.S);
}
template <int D>
__device__ __forceinline__ void ac_apply(const AffCov<D>& e, const double (&P)[D][D], double (&out)[D][D]) {
    double T1[D][D];
    ce_mul<D, false, false>(e.M, P, T1);
    ce_mul<D, false, true>(T1, e.M, out);
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) out[i][j] += e.S[i][j];
    ce_sym<D>(out);
}
template <int D>
__device__ __forceinline__ void ac_scan(AffCov<D>& pre, int lane) {      // inclusive, lane order = order of application
#pragma unroll 1
    for (int k = 0; k < 6; ++k) {
        const int off = 1 << k;
        AffCov<D> other, res;
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) {
                other.M[i][j] = __shfl_up(pre.M[i][j], off);
                other.S[i][j] = __shfl_up(pre.S[i][j], off);
            }
        ac_combine<D>(other, pre, res);
        if (lane >= off) pre = res;
    }
}

template <int D>
__device__ void setup_side_scan(const ModelDev& m, const Tab& tb, long long T) {
    constexpr int DD = D * D;
    constexpr double kTolFine = 16.0 * kTol, kTolCoarse = 256.0 * kTol;
    const int lane = threadIdx.x & 63;
    const int th = (int)tb.hdr[1], n0 = (int)tb.hdr[5];
    const bool notfirst = (tb.hdr[6] & 1) != 0, notlast = (tb.hdr[6] & 2) != 0;
    const int nh = th * kTile;
    double hv[D];
#pragma unroll
    for (int k = 0; k < D; ++k) hv[k] = m.H[k];
    if (lane == 0) tb.misc[13] = (double)wall_clock64();
    auto quad = [&](const double (&Pm)[D][D]) {     // H Symmetric(P) H'
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < D; ++c) {
            double v = 0.0;
#pragma unroll
            for (int r = 0; r < D; ++r) v = fma(hv[r], (r <= c ? Pm[r][c] : Pm[c][r]), v);
            s = fma(v, hv[c], s);
        }
        return s;
    };
    // ---- (c) the tail: Ps behind the last step is the stationary filtered covariance; backwards with the stationary (G, L)
    double base[D][D];
    AffCov<D> pre;
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const int r = i < j ? i : j, c = i < j ? j : i;
            base[i][j] = tb.s_Pf[(size_t)n0 * DD + r * D + c];
            pre.M[i][j] = tb.h_G[(size_t)n0 * DD + i * D + j];      // (the stationary gain: the constant block is still being written)
            pre.S[i][j] = tb.s_L[(size_t)n0 * DD + r * D + c];
        }
    ac_scan<D>(pre, lane);                  // lane j: (G^(j+1), sum_{i <= j} G^i L G^i')
    if (lane == 0) tb.t_vb[0] = quad(base);
    int n1 = -1;
    double Pinf[D][D];
#pragma unroll 1
    for (int b = 0; b * 64 < kTailMax; ++b) {
        double Ps[D][D], prev[D][D];
        ac_apply<D>(pre, base, Ps);                                 // Ps at distance 64 b + lane + 1 from the end
        bool moved = false, moved_fine = false;
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const double up = __shfl_up(Ps[i][j], 1);
                prev[i][j] = lane == 0 ? base[i][j] : up;
            }
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const double dlt = fabs(Ps[i][j] - prev[i][j]), sc = 0.5 * (fabs(prev[i][i]) + fabs(prev[j][j]));
                moved = moved || !(dlt <= kTolCoarse * sc);
                moved_fine = moved_fine || !(dlt <= kTolFine * sc);
            }
        const unsigned long long still_fine = __ballot(!moved_fine), still = __ballot(!moved);
        const int lstar = still_fine ? __ffsll((long long)still_fine) - 1 : (still ? 63 : 64);
        const int k = b * 64 + lane + 1;
        if (k < kTailMax && lane <= lstar) tb.t_vb[k] = quad(Ps);
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) base[i][j] = __shfl(Ps[i][j], 63);       // next block's start; the most converged value of this one
        if (lstar < 64) {
            n1 = b * 64 + lstar + 1;
            break;
        }
    }
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) Pinf[i][j] = base[i][j];
    const bool applies = n1 >= 0 && n1 < kTailMax && (long long)nh + (notlast ? 0 : n1) + 1 <= T;
    if (lane == 0) {
        tb.hdr[3] = notlast ? 0 : n1;
        if (!applies) tb.hdr[0] = 0;
    }
    if (!applies) return;
    if (lane == 0) tb.misc[14] = (double)wall_clock64();
    const double vb_ss = quad(Pinf);
    if (lane == 0) {
        tb.ssc[SS<D>::vb] = vb_ss;
        tb.h_vb[n0] = vb_ss;
    }
    // ---- (d) the head: from Ps_{n0} = stationary backwards through the steps' own (G_t, L_t); lane l of a block holds step hi - l
    if (!notfirst) {
#pragma unroll 1
        for (int hi = n0; hi >= 1; hi -= 64) {
            const int t = hi - lane;
            AffCov<D> el;
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    el.M[i][j] = t >= 1 ? tb.h_G[(size_t)t * DD + i * D + j] : (i == j ? 1.0 : 0.0);
                    el.S[i][j] = t >= 1 ? tb.s_L[(size_t)t * DD + (i < j ? i : j) * D + (i < j ? j : i)] : 0.0;
                }
            ac_scan<D>(el, lane);
            double Ps[D][D];
            ac_apply<D>(el, Pinf, Ps);                              // Ps_{t-1}
            if (t >= 1) tb.h_vb[t - 1] = quad(Ps);
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) Pinf[i][j] = __shfl(Ps[i][j], 63);
        }
    }
    if (lane == 0) tb.misc[15] = (double)wall_clock64();
}

// =================================================================================================================================
// tile passes
// =================================================================================================================================
template <int D>
struct Coef {
    double A[D][D], a[D], h[D], hh, kA[D], rS, G[D][D], c[D];
    __device__ __forceinline__ void load(const double* __restrict__ ss) {
#pragma unroll
        for (int i = 0; i < D; ++i) {
#pragma unroll
            for (int k = 0; k < D; ++k) {
                A[i][k] = ss[SS<D>::A + i * D + k];
                G[i][k] = ss[SS<D>::G + i * D + k];
            }
            a[i] = ss[SS<D>::a + i];
            h[i] = ss[SS<D>::h + i];
            kA[i] = ss[SS<D>::kA + i];
            c[i] = ss[SS<D>::c + i];
        }
        hh = ss[SS<D>::hh];
        rS = ss[SS<D>::rS];
    }
};

__device__ __forceinline__ void load8(const double* __restrict__ p, long long t0, long long T, double (&v)[kSub]) {
    if (t0 + kSub <= T) {
        if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
            const double2* q = reinterpret_cast<const double2*>(p + t0);
#pragma unroll
            for (int j = 0; j < kSub / 2; ++j) {
                const double2 w = q[j];
                v[2 * j] = w.x;
                v[2 * j + 1] = w.y;
            }
        } else {
#pragma unroll
            for (int j = 0; j < kSub; ++j) v[j] = p[t0 + j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < kSub; ++j) v[j] = (t0 + j < T) ? p[t0 + j] : 0.0;
    }
}

__device__ __forceinline__ void store8(double* __restrict__ p, long long t0, long long T, const double (&v)[kSub]) {
    if (t0 + kSub <= T) {
        if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
            double2* q = reinterpret_cast<double2*>(p + t0);
#pragma unroll
            for (int j = 0; j < kSub / 2; ++j) q[j] = make_double2(v[2 * j], v[2 * j + 1]);
        } else {
#pragma unroll
            for (int j = 0; j < kSub; ++j) p[t0 + j] = v[j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < kSub; ++j)
            if (t0 + j < T) p[t0 + j] = v[j];
    }
}

// A wave's tile of 512 consecutive output values, eight per lane, leaves through the wave's own LDS row (kTileRow doubles): lane l's pairs go
// in at 16-byte slots 4 l + l / 4 + j / 2 as the backward walk produces them (no array of eight held in registers; the pad keeps the
// 128-bit writes of sixteen lanes on distinct banks) and come out transposed, so that every store instruction writes 1 KB of consecutive
// bytes -- eight whole lines -- instead of 64 sixteen-byte pieces 64 bytes apart that the L2 has to merge (pass 2 at T = 1e7: 75 -> 61 us
// at d = 3, 69 -> 47 us at d = 2).  Ragged tiles and pointers off a 16-byte boundary store each lane's own slots, element by element.
constexpr int kTileRow = kTile + 32;
typedef double v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int tile_slot(int lane) { return lane * 4 + (lane >> 2); }
__device__ __forceinline__ void flush_tile(double* __restrict__ p, long long tile_t0, long long T, const v2d* r2, int lane) {
    if (tile_t0 + kTile <= T && (reinterpret_cast<uintptr_t>(p) & 15) == 0) {      // (wave-uniform)
        v2d* q = reinterpret_cast<v2d*>(p + tile_t0);
#pragma unroll
        for (int k = 0; k < kSub / 2; ++k) {
            const int e = k * 64 + lane, ls = e >> 2;
            q[e] = r2[ls * 4 + (ls >> 2) + (e & 3)];
        }
    } else {
        const long long t0 = tile_t0 + lane * kSub;
        const int wb = tile_slot(lane);
#pragma unroll
        for (int j = 0; j < kSub / 2; ++j) {
            const v2d w = r2[wb + j];
            if (t0 + 2 * j < T) p[t0 + 2 * j] = w.x;
            if (t0 + 2 * j + 1 < T) p[t0 + 2 * j + 1] = w.y;
        }
    }
}

// two consecutive values at t (even offset inside the lane's eight): one 16-byte store where the pointer allows
__device__ __forceinline__ void store2(double* __restrict__ p, long long t, long long T, double a, double b) {
    if (t + 1 < T && (reinterpret_cast<uintptr_t>(p + t) & 15) == 0) {
        *reinterpret_cast<double2*>(p + t) = make_double2(a, b);
    } else {
        if (t < T) p[t] = a;
        if (t + 1 < T) p[t + 1] = b;
    }
}

// forward half of a stationary tile: local recursion from `mu` (lane 0: the tile's carry, others: zero), wave scan with Phi^(8 2^k),
// second local recursion from the scanned start -> the innovations r[8].  Returns the lane's inclusive scan value (lane 63: tile end).
template <int D, bool KEEP = false>
__device__ __forceinline__ void tile_forward(const Coef<D>& cf, const double* __restrict__ pw_phi, const double (&y)[kSub], int nvalid,
                                             int lane, const double (&mu_in)[D], double (&r)[kSub], double (&fend)[D],
                                             double (*mus)[D] = nullptr /* KEEP: the predicted mean before each of the lane's steps */) {
    constexpr int DD = D * D;
    double mu[D];
#pragma unroll
    for (int i = 0; i < D; ++i) mu[i] = mu_in[i];
#pragma unroll
    for (int j = 0; j < kSub; ++j) {
        double rr = y[j] - cf.hh;
#pragma unroll
        for (int k = 0; k < D; ++k) rr = fma(-cf.h[k], mu[k], rr);
        rr = (j < nvalid) ? rr : 0.0;
        r[j] = rr;
        double nm[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double v = fma(cf.kA[i], rr, cf.a[i]);
#pragma unroll
            for (int k = 0; k < D; ++k) v = fma(cf.A[i][k], mu[k], v);
            nm[i] = v;
        }
#pragma unroll
        for (int i = 0; i < D; ++i) mu[i] = nm[i];
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int off = 1 << k;
        double g[D];
#pragma unroll
        for (int i = 0; i < D; ++i) g[i] = __shfl_up(mu[i], off);
        const double* __restrict__ M = pw_phi + (size_t)(3 + k) * DD;
        if (lane >= off) {
#pragma unroll
            for (int i = 0; i < D; ++i) {
                double v = mu[i];
#pragma unroll
                for (int l = 0; l < D; ++l) v = fma(M[i * D + l], g[l], v);
                mu[i] = v;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < D; ++i) fend[i] = mu[i];
    // start state of this lane: the inclusive value of the lane before it (lane 0: the carry)
    double st[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const double v = __shfl_up(mu[i], 1);
        st[i] = (lane == 0) ? mu_in[i] : v;
    }
    if constexpr (!KEEP) {
        // the first recursion's innovations r0 were computed from a zero start (lane 0: from the carry itself); the start state moves
        // step j's predicted mean by Phi^j st, its innovation by -h' Phi^j st: eight dot products instead of a second recursion
        const double* __restrict__ W = pw_phi - CL<D>::pphi + CL<D>::WJ;      // (pw_phi = cst + CL::pphi)
#pragma unroll
        for (int j = 0; j < kSub; ++j) {
            double rr = r[j];
#pragma unroll
            for (int k = 0; k < D; ++k) rr = fma(-W[j * D + k], (lane == 0) ? 0.0 : st[k], rr);
            r[j] = (j < nvalid) ? rr : 0.0;
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < kSub; ++j) {
        double rr = y[j] - cf.hh;
#pragma unroll
        for (int k = 0; k < D; ++k) rr = fma(-cf.h[k], st[k], rr);
        rr = (j < nvalid) ? rr : 0.0;
        r[j] = rr;
        if (KEEP) {
#pragma unroll
            for (int k = 0; k < D; ++k) mus[j][k] = st[k];
        }
        double nm[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double v = fma(cf.kA[i], rr, cf.a[i]);
#pragma unroll
            for (int k = 0; k < D; ++k) v = fma(cf.A[i][k], st[k], v);
            nm[i] = v;
        }
#pragma unroll
        for (int i = 0; i < D; ++i) st[i] = nm[i];
    }
}

// backward half: local recursion lam <- G lam + c r from `lam_in` (lane 63: the tile's carry, others zero), reverse wave scan.
// Returns the lane's inclusive value (lane 0: lam after the tile's first step) and the lane's start value.
template <int D>
__device__ __forceinline__ void tile_backward(const Coef<D>& cf, const double* __restrict__ pw_g, const double (&r)[kSub], int lane,
                                              const double (&lam_in)[D], double (&bend)[D], double (&lstart)[D]) {
    constexpr int DD = D * D;
    double lam[D];
#pragma unroll
    for (int i = 0; i < D; ++i) lam[i] = lam_in[i];
#pragma unroll
    for (int j = kSub - 1; j >= 0; --j) {
        double nl[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double v = cf.c[i] * r[j];
#pragma unroll
            for (int k = 0; k < D; ++k) v = fma(cf.G[i][k], lam[k], v);
            nl[i] = v;
        }
#pragma unroll
        for (int i = 0; i < D; ++i) lam[i] = nl[i];
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int off = 1 << k;
        double g[D];
#pragma unroll
        for (int i = 0; i < D; ++i) g[i] = __shfl_down(lam[i], off);
        const double* __restrict__ M = pw_g + (size_t)(3 + k) * DD;
        if (lane + off < 64) {
#pragma unroll
            for (int i = 0; i < D; ++i) {
                double v = lam[i];
#pragma unroll
                for (int l = 0; l < D; ++l) v = fma(M[i * D + l], g[l], v);
                lam[i] = v;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < D; ++i) {
        bend[i] = lam[i];
        const double v = __shfl_down(lam[i], 1);
        lstart[i] = (lane == 63) ? lam_in[i] : v;
    }
}

// ---- pass 1's form of the two halves: a tile is HALF a wave -- 32 lanes of sixteen steps -- with zero carries.  The in-tile scans cost the
// same per level whatever a lane holds, so sixteen steps per lane halve them per step (five levels with Phi^(16 2^k) instead of six with
// Phi^(8 2^k)); pass 1 keeps nothing but y and r per lane, so the registers are there (pass 2, with its outputs, stays at eight).  The two
// halves of a wave are two tiles: the scans never cross lane 32.
constexpr int kSub2 = 2 * kSub;
template <int D>
__device__ __forceinline__ void half_tile_forward(const Coef<D>& cf, const double* __restrict__ pw_phi, const double (&y)[kSub2], int nvalid, int sl,
                                                  double (&r)[kSub2], double (&fend)[D]) {
    constexpr int DD = D * D;
    double mu[D];
#pragma unroll
    for (int i = 0; i < D; ++i) mu[i] = 0.0;
#pragma unroll
    for (int j = 0; j < kSub2; ++j) {
        double rr = y[j] - cf.hh;
#pragma unroll
        for (int k = 0; k < D; ++k) rr = fma(-cf.h[k], mu[k], rr);
        rr = (j < nvalid) ? rr : 0.0;
        r[j] = rr;
        double nm[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double v = fma(cf.kA[i], rr, cf.a[i]);
#pragma unroll
            for (int k = 0; k < D; ++k) v = fma(cf.A[i][k], mu[k], v);
            nm[i] = v;
        }
#pragma unroll
        for (int i = 0; i < D; ++i) mu[i] = nm[i];
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int off = 1 << k;
        double g[D];
#pragma unroll
        for (int i = 0; i < D; ++i) g[i] = __shfl_up(mu[i], off);
        const double* __restrict__ M = pw_phi + (size_t)(4 + k) * DD;
        if (sl >= off) {
#pragma unroll
            for (int i = 0; i < D; ++i) {
                double v = mu[i];
#pragma unroll
                for (int l = 0; l < D; ++l) v = fma(M[i * D + l], g[l], v);
                mu[i] = v;
            }
        }
    }
    double st[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        fend[i] = mu[i];
        const double v = __shfl_up(mu[i], 1);
        st[i] = (sl == 0) ? 0.0 : v;
    }
    const double* __restrict__ W = pw_phi - CL<D>::pphi + CL<D>::WJ;
#pragma unroll
    for (int j = 0; j < kSub2; ++j) {
        double rr = r[j];
#pragma unroll
        for (int k = 0; k < D; ++k) rr = fma(-W[j * D + k], st[k], rr);
        r[j] = (j < nvalid) ? rr : 0.0;
    }
}
template <int D>
__device__ __forceinline__ void half_tile_backward(const Coef<D>& cf, const double* __restrict__ pw_g, const double (&r)[kSub2], int sl, double (&bend)[D]) {
    constexpr int DD = D * D;
    double lam[D];
#pragma unroll
    for (int i = 0; i < D; ++i) lam[i] = 0.0;
#pragma unroll
    for (int j = kSub2 - 1; j >= 0; --j) {
        double nl[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double v = cf.c[i] * r[j];
#pragma unroll
            for (int k = 0; k < D; ++k) v = fma(cf.G[i][k], lam[k], v);
            nl[i] = v;
        }
#pragma unroll
        for (int i = 0; i < D; ++i) lam[i] = nl[i];
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int off = 1 << k;
        double g[D];
#pragma unroll
        for (int i = 0; i < D; ++i) g[i] = __shfl_down(lam[i], off);
        const double* __restrict__ M = pw_g + (size_t)(4 + k) * DD;
        if (sl + off < 32) {
#pragma unroll
            for (int i = 0; i < D; ++i) {
                double v = lam[i];
#pragma unroll
                for (int l = 0; l < D; ++l) v = fma(M[i * D + l], g[l], v);
                lam[i] = v;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < D; ++i) bend[i] = lam[i];
}

// ---- head tiles: per-step gains from the tables, general affine wave scan (matrix, vector) -------------------------------------------
// One wave walks the th head tiles in order.  Forward: innovations of the head -> tb.h_r, carry into the first stationary workgroup -> MUb[0],
// sum r^2 / S -> misc[0].
template <int D>
__device__ void head_forward(const Tab& tb, const double* __restrict__ y, long long T, int lane) {
    const int th = (int)tb.hdr[1];
    const long long n0c = tb.hdr[2];
    auto tix = [&](long long t) { return t < n0c ? t : n0c; };       // the per-step tables end with the stationary step n0
    const double* __restrict__ ss = tb.ssc;
    double A[D][D], a[D], h[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
#pragma unroll
        for (int k = 0; k < D; ++k) A[i][k] = ss[SS<D>::A + i * D + k];
        a[i] = ss[SS<D>::a + i];
        h[i] = ss[SS<D>::h + i];
    }
    const double hh = ss[SS<D>::hh];
    double carry[D];
#pragma unroll
    for (int i = 0; i < D; ++i) carry[i] = ss[SS<D>::mu0 + i];
    double acc = 0.0;
    for (int tile = 0; tile < th; ++tile) {
        const long long t0 = (long long)tile * kTile + lane * kSub;
        double yv[kSub];
        load8(y, t0, T, yv);
        double M[D][D], mu[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            mu[i] = (lane == 0) ? carry[i] : 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) M[i][k] = (i == k) ? 1.0 : 0.0;
        }
#pragma unroll
        for (int j = 0; j < kSub; ++j) {
            double kA[D];
#pragma unroll
            for (int i = 0; i < D; ++i) kA[i] = tb.h_kA[tix(t0 + j) * D + i];
            double rr = yv[j] - hh;
#pragma unroll
            for (int k = 0; k < D; ++k) rr = fma(-h[k], mu[k], rr);
            double nm[D], hM[D], nM[D][D];
#pragma unroll
            for (int c = 0; c < D; ++c) {
                double v = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) v = fma(h[k], M[k][c], v);
                hM[c] = v;
            }
#pragma unroll
            for (int i = 0; i < D; ++i) {
                double v = fma(kA[i], rr, a[i]);
#pragma unroll
                for (int k = 0; k < D; ++k) v = fma(A[i][k], mu[k], v);
                nm[i] = v;
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    double w = -kA[i] * hM[c];
#pragma unroll
                    for (int k = 0; k < D; ++k) w = fma(A[i][k], M[k][c], w);
                    nM[i][c] = w;               // (A - kA h') M
                }
            }
#pragma unroll
            for (int i = 0; i < D; ++i) {
                mu[i] = nm[i];
#pragma unroll
                for (int c = 0; c < D; ++c) M[i][c] = nM[i][c];
            }
        }
        // inclusive scan of (M, mu): later o earlier = (M_l M_e, M_l mu_e + mu_l)
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int off = 1 << k;
            double g[D], Mg[D][D];
#pragma unroll
            for (int i = 0; i < D; ++i) {
                g[i] = __shfl_up(mu[i], off);
#pragma unroll
                for (int c = 0; c < D; ++c) Mg[i][c] = __shfl_up(M[i][c], off);
            }
            if (lane >= off) {
                double nM[D][D];
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    double v = mu[i];
#pragma unroll
                    for (int l = 0; l < D; ++l) v = fma(M[i][l], g[l], v);
                    mu[i] = v;
#pragma unroll
                    for (int c = 0; c < D; ++c) {
                        double w = 0.0;
#pragma unroll
                        for (int l = 0; l < D; ++l) w = fma(M[i][l], Mg[l][c], w);
                        nM[i][c] = w;
                    }
                }
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int c = 0; c < D; ++c) M[i][c] = nM[i][c];
            }
        }
        double st[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            const double v = __shfl_up(mu[i], 1);
            st[i] = (lane == 0) ? carry[i] : v;
            carry[i] = __shfl(mu[i], 63);
        }
#pragma unroll
        for (int j = 0; j < kSub; ++j) {
            double rr = yv[j] - hh;
#pragma unroll
            for (int k = 0; k < D; ++k) rr = fma(-h[k], st[k], rr);
            tb.h_r[t0 + j] = rr;
            acc = fma(rr * rr, tb.h_iS[tix(t0 + j)], acc);
            double nm[D];
#pragma unroll
            for (int i = 0; i < D; ++i) {
                double v = fma(tb.h_kA[tix(t0 + j) * D + i], rr, a[i]);
#pragma unroll
                for (int k = 0; k < D; ++k) v = fma(A[i][k], st[k], v);
                nm[i] = v;
            }
#pragma unroll
            for (int i = 0; i < D; ++i) st[i] = nm[i];
        }
    }
    // fixed-order wave sum
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_down(acc, off);
    if (lane == 0) {
        tb.misc[0] = acc;
#pragma unroll
        for (int i = 0; i < D; ++i) tb.MUb[i] = carry[i];
    }
}

// Backward over the head tiles from LAMb[0]: posterior marginals of the head.
template <int D>
__device__ void head_backward(const Tab& tb, const double* __restrict__ y, const double* __restrict__ Rnew, int rnew_per_step,
                              double* __restrict__ mean, double* __restrict__ var, long long T, int lane) {
    constexpr int DD = D * D;
    const int th = (int)tb.hdr[1];
    const long long n0c = tb.hdr[2];
    auto tix = [&](long long t) { return t < n0c ? t : n0c; };
    const double* __restrict__ ss = tb.ssc;
    double h[D];
#pragma unroll
    for (int i = 0; i < D; ++i) h[i] = ss[SS<D>::h + i];
    double carry[D];
#pragma unroll
    for (int i = 0; i < D; ++i) carry[i] = tb.LAMb[i];
    const double rn0 = Rnew[0];
    for (int tile = th - 1; tile >= 0; --tile) {
        const long long t0 = (long long)tile * kTile + lane * kSub;
        double yv[kSub], rv[kSub];
        load8(y, t0, T, yv);
        load8(tb.h_r, t0, T, rv);
        double M[D][D], lam[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            lam[i] = (lane == 63) ? carry[i] : 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) M[i][k] = (i == k) ? 1.0 : 0.0;
        }
#pragma unroll
        for (int j = kSub - 1; j >= 0; --j) {
            const double* __restrict__ Gt = tb.h_G + (size_t)tix(t0 + j) * DD;
            double nl[D], nM[D][D];
#pragma unroll
            for (int i = 0; i < D; ++i) {
                double v = tb.h_c[tix(t0 + j) * D + i] * rv[j];
#pragma unroll
                for (int k = 0; k < D; ++k) v = fma(Gt[i * D + k], lam[k], v);
                nl[i] = v;
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    double w = 0.0;
#pragma unroll
                    for (int k = 0; k < D; ++k) w = fma(Gt[i * D + k], M[k][c], w);
                    nM[i][c] = w;
                }
            }
#pragma unroll
            for (int i = 0; i < D; ++i) {
                lam[i] = nl[i];
#pragma unroll
                for (int c = 0; c < D; ++c) M[i][c] = nM[i][c];
            }
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int off = 1 << k;
            double g[D], Mg[D][D];
#pragma unroll
            for (int i = 0; i < D; ++i) {
                g[i] = __shfl_down(lam[i], off);
#pragma unroll
                for (int c = 0; c < D; ++c) Mg[i][c] = __shfl_down(M[i][c], off);
            }
            if (lane + off < 64) {
                double nM[D][D];
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    double v = lam[i];
#pragma unroll
                    for (int l = 0; l < D; ++l) v = fma(M[i][l], g[l], v);
                    lam[i] = v;
#pragma unroll
                    for (int c = 0; c < D; ++c) {
                        double w = 0.0;
#pragma unroll
                        for (int l = 0; l < D; ++l) w = fma(M[i][l], Mg[l][c], w);
                        nM[i][c] = w;
                    }
                }
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int c = 0; c < D; ++c) M[i][c] = nM[i][c];
            }
        }
        double st[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            const double v = __shfl_down(lam[i], 1);
            st[i] = (lane == 63) ? carry[i] : v;
            carry[i] = __shfl(lam[i], 0);
        }
        double mo[kSub], vo[kSub];
        if (rnew_per_step) load8(Rnew, t0, T, vo);
#pragma unroll
        for (int j = kSub - 1; j >= 0; --j) {
            const double* __restrict__ Gt = tb.h_G + (size_t)tix(t0 + j) * DD;
            double m = fma(-tb.h_rS[tix(t0 + j)], rv[j], yv[j]);
#pragma unroll
            for (int k = 0; k < D; ++k) m = fma(h[k], st[k], m);
            mo[j] = m;
            vo[j] = tb.h_vb[tix(t0 + j)] + (rnew_per_step ? vo[j] : rn0);
            double nl[D];
#pragma unroll
            for (int i = 0; i < D; ++i) {
                double v = tb.h_c[tix(t0 + j) * D + i] * rv[j];
#pragma unroll
                for (int k = 0; k < D; ++k) v = fma(Gt[i * D + k], st[k], v);
                nl[i] = v;
            }
#pragma unroll
            for (int i = 0; i < D; ++i) st[i] = nl[i];
        }
        store8(mean, t0, T, mo);
        store8(var, t0, T, vo);
    }
}

template <int D>
__device__ __forceinline__ void matvec_acc(const double* __restrict__ M, const double (&x)[D], double (&y)[D]) {     // y += M x
#pragma unroll
    for (int i = 0; i < D; ++i) {
        double v = y[i];
#pragma unroll
        for (int l = 0; l < D; ++l) v = fma(M[i * D + l], x[l], v);
        y[i] = v;
    }
}

// Carries of the kBlk tiles of a workgroup from the workgroup's own (mu at its first step, lam behind its last step) and the tiles'
// zero-carry elements F, B0: mu_{w+1} = Phi^512 mu_w + F_w; lam_w (behind tile w) = G^512 lam_{w+1} + B0_{w+1} - C_{w+1} mu_{w+1}, where
// the coupling C is B_512 for a full tile, Blt for the last (ragged) tile of the series and nothing for a tile beyond the end.
// Run by ONE thread; sMu [kBlk + 1][D] and sLam [kBlk][D] live in LDS (a handful of wave-uniform values would otherwise occupy a
// vector register each in all 64 lanes).  Returns mu behind the last tile in sMu[kBlk] and lam in front of the first in lam_out.
template <int D, class FGet, class BGet>
__device__ __forceinline__ void block_carries(const double* __restrict__ cst, FGet F, BGet B0, long long tile0, long long ntiles,
                                              const double (&mu_in)[D], const double (&lam_in)[D], double (*sMu)[D], double (*sLam)[D],
                                              double (&lam_out)[D]) {
    constexpr int DD = D * D;
    const double* __restrict__ M = cst + CL<D>::pphi + kLogTile * DD;
    const double* __restrict__ G = cst + CL<D>::pg + kLogTile * DD;
    double mu[D];
#pragma unroll
    for (int i = 0; i < D; ++i) mu[i] = sMu[0][i] = mu_in[i];
#pragma unroll
    for (int w = 0; w < kBlk; ++w) {
        double n[D];
#pragma unroll
        for (int i = 0; i < D; ++i) n[i] = F(w, i);
        matvec_acc<D>(M, mu, n);
#pragma unroll
        for (int i = 0; i < D; ++i) mu[i] = sMu[w + 1][i] = n[i];
    }
    double l[D];
#pragma unroll
    for (int i = 0; i < D; ++i) l[i] = lam_in[i];
#pragma unroll
    for (int w = kBlk - 1; w >= 0; --w) {
        double n[D], m[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            sLam[w][i] = l[i];          // behind tile w
            n[i] = B0(w, i);
            m[i] = sMu[w][i];
        }
        const long long tile = tile0 + w;
        if (tile >= ntiles) continue;          // (beyond the end: nothing to absorb -- a time shard's lam must not be advanced across it)
        {
            const double* __restrict__ C = cst + (tile == ntiles - 1 ? CL<D>::Blt : CL<D>::B512);
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int k = 0; k < D; ++k) n[i] = fma(-C[i * D + k], m[k], n[i]);
        }
        matvec_acc<D>(G, l, n);
#pragma unroll
        for (int i = 0; i < D; ++i) l[i] = n[i];
    }
#pragma unroll
    for (int i = 0; i < D; ++i) lam_out[i] = l[i];
}

// pass 1: the stationary tiles with zero carries -> per-tile elements F, B0 and the workgroup's element (Fb, B0b)
template <int D, bool POST>
__global__ __launch_bounds__(kBlkThreads / 2) void k_reduce(const long long* __restrict__ hdr, const double* __restrict__ cst, const double* __restrict__ y,
                                                double* __restrict__ F, double* __restrict__ B0, double* __restrict__ Fb,
                                                double* __restrict__ B0b, long long T, long long ntiles, ModelDev m, Tab tb, int side) {
    if (hdr[0] == 0) return;
    if (POST && blockIdx.x == 0) {      // the extra workgroup (dispatched first): variance tables for pass 2 (not for adjoint calls)
        // (d <= kCovScanMaxD: the tables were built by the set-up kernel's second wave, unless the sequential recursions were asked for)
        if (side && threadIdx.x < 64 && (D > kCovScanMaxD || (tb.hdr[6] & 4) != 0)) setup_side<D>(m, tb, T);
        return;
    }
    const long long wg = (long long)blockIdx.x - (POST ? 1 : 0);
    __shared__ double sF[kBlk][D], sB0[kBlk][D], sMu[kBlk + 1][D], sLam[kBlk][D];
    // (kBlkThreads / 2 threads: four waves, a wave's two halves are two tiles of 32 lanes x 16 steps)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sl = lane & 31, wt = 2 * wave + (lane >> 5);
    const long long tile0 = hdr[1] + wg * kBlk;
    if (tile0 >= ntiles) return;            // (the launch is sized for one head tile)
    const long long tile = tile0 + wt;
    double fend[D], bend[D];
#pragma unroll
    for (int i = 0; i < D; ++i) fend[i] = bend[i] = 0.0;
    {
        // (both halves run the same instructions; a half whose tile lies behind the series' end has no valid step)
        Coef<D> cf;
        cf.load(cst);
        const long long t0 = tile * kTile + sl * kSub2;
        double yv[kSub2], r[kSub2], ya[kSub], yb[kSub];
        load8(y, t0, T, ya);
        load8(y, t0 + kSub, T, yb);
#pragma unroll
        for (int j = 0; j < kSub; ++j) {
            yv[j] = ya[j];
            yv[kSub + j] = yb[j];
        }
        const long long left = T - t0;
        const int nvalid = (tile < ntiles) ? (left >= kSub2 ? kSub2 : (left > 0 ? (int)left : 0)) : 0;
        half_tile_forward<D>(cf, cst + CL<D>::pphi, yv, nvalid, sl, r, fend);
        if (POST) half_tile_backward<D>(cf, cst + CL<D>::pg, r, sl, bend);
    }
    if (sl == 31) {
#pragma unroll
        for (int i = 0; i < D; ++i) {
            sF[wt][i] = fend[i];
            if (tile < ntiles) F[tile * D + i] = fend[i];
        }
    }
    if (POST && sl == 0) {
#pragma unroll
        for (int i = 0; i < D; ++i) {
            sB0[wt][i] = bend[i];
            if (tile < ntiles) B0[tile * D + i] = bend[i];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double zero[D], lout[D];
#pragma unroll
        for (int i = 0; i < D; ++i) zero[i] = 0.0;
        block_carries<D>(cst, [&](int w, int i) { return sF[w][i]; }, [&](int w, int i) { return POST ? sB0[w][i] : 0.0; }, tile0, ntiles, zero,
                         zero, sMu, sLam, lout);
#pragma unroll
        for (int i = 0; i < D; ++i) {
            Fb[wg * D + i] = sMu[kBlk][i];
            if (POST) B0b[wg * D + i] = lout[i];
        }
    }
    __syncthreads();
    if (threadIdx.x < kBlk * D) {          // the tiles' carries under zero workgroup carries, for pass 2's closed form
        const int w = threadIdx.x / D, i = threadIdx.x % D;
        tb.Pw[(tile0 + w) * D + i] = sMu[w][i];
        if (POST) tb.Lw[(tile0 + w) * D + i] = sLam[w][i];
    }
}

// carries of the workgroups: MUb[b+1] = Phi^4096 MUb[b] + Fb[b] from MUb[0] (head_forward); LAMb[b] = G^4096 LAMb[b+1] + B0b[b] - C MUb[b]
// from LAMb[nblk] = 0 (C = B_4096, the last workgroup: Blb).  One block of 512 lanes, kQ consecutive elements per lane.  A series of
// one slice (512 kQ workgroups: T <= 1.6e7 at d <= 6) makes ONE round trip to memory -- Fb and B0b are fetched together at the start and
// the carries never leave the registers between the two directions -- plus the two scans: Hillis-Steele over the lanes with the powers
// 2^(12 + log2 kQ + k) from LDS, through two alternating buffers (one barrier per round).  Longer series walk the slices, chained through
// the slice's end state, and read the elements again on the way back.
template <int D, bool POST>
__global__ __launch_bounds__(512) void k_carry(const long long* __restrict__ hdr, const double* __restrict__ cst, const double* __restrict__ Fb,
                                                const double* __restrict__ B0b, double* MUb, double* LAMb, long long ntiles,
                                                int lam_given /* time shards: LAMb[N] holds the lam behind the segment (else zero) */) {
    if (hdr[0] == 0) return;
    constexpr int DD = D * D;
    constexpr int kLanes = 512, kRounds = 9;
    if (hdr[7] != 0) {
        // short memory (k_setup_core): mu into workgroup b is the element of workgroup b - 1, lam in front of workgroup b its own element
        // coupled to that mu -- plus, for the series' last workgroup only, what the boundary lam leaves after its (few) steps
        const long long N = nblk_max_for(hdr[1], ntiles);
        for (long long b = (long long)blockIdx.x * kLanes + threadIdx.x; b < N; b += (long long)gridDim.x * kLanes) {
            double mu[D];
#pragma unroll
            for (int i = 0; i < D; ++i) mu[i] = (b == 0) ? MUb[i] : Fb[(b - 1) * D + i];
            if (b > 0) {
#pragma unroll
                for (int i = 0; i < D; ++i) MUb[b * D + i] = mu[i];
            }
            if (POST) {
                const bool last = b == N - 1;
                const double* __restrict__ C = cst + (last ? CL<D>::Blb : CL<D>::Bblk);
                double l[D];
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    double v = B0b[b * D + i];
#pragma unroll
                    for (int k = 0; k < D; ++k) v = fma(-C[i * D + k], mu[k], v);
                    l[i] = v;
                }
                if (last && lam_given) {
                    double lin[D];
#pragma unroll
                    for (int i = 0; i < D; ++i) lin[i] = LAMb[N * D + i];
                    matvec_acc<D>(cst + CL<D>::GLb, lin, l);
                }
#pragma unroll
                for (int i = 0; i < D; ++i) LAMb[b * D + i] = l[i];
            }
        }
        if (POST && !lam_given && blockIdx.x == 0 && threadIdx.x < D) LAMb[N * D + threadIdx.x] = 0.0;
        return;
    }
    if (blockIdx.x != 0) return;      // (the scans below are one workgroup's; the other workgroups of the launch exist for the short-memory path)
    constexpr int lQ = D <= 6 ? 3 : 2, kQ = 1 << lQ;
    __shared__ double sv[2][D][kLanes];
    __shared__ double sPw[2][kRounds][DD];      // the scans' matrices, fetched once (a scalar load per round would cost a memory round trip each)
    const int tid = threadIdx.x;
    const long long N = nblk_max_for(hdr[1], ntiles);
    const long long slice = (long long)kLanes * kQ;
    const long long nslices = (N + slice - 1) / slice;
    const bool one = nslices == 1;
    for (int idx = tid; idx < 2 * kRounds * DD; idx += kLanes) {
        const int w = idx / (kRounds * DD), r = idx % (kRounds * DD);
        sPw[w][r / DD][r % DD] = cst[(w == 0 ? CL<D>::pphi : CL<D>::pg) + (kLogBlk + lQ) * DD + r];
    }
    auto step = [&](const double* __restrict__ M, double (&s)[D], const double (&el)[D]) {
        double n[D];
#pragma unroll
        for (int i = 0; i < D; ++i) n[i] = el[i];
        matvec_acc<D>(M, s, n);
#pragma unroll
        for (int i = 0; i < D; ++i) s[i] = n[i];
    };
    // inclusive scan over the lanes; the result is also left in sv[1] (kRounds is odd: the last round reads sv[0])
    auto block_scan = [&](double (&s)[D], int which, bool up) {
#pragma unroll 1
        for (int k = 0; k < kRounds; ++k) {
            const int off = 1 << k;
#pragma unroll
            for (int i = 0; i < D; ++i) sv[k & 1][i][tid] = s[i];
            __syncthreads();
            const int src = up ? tid - off : tid + off;
            if (src >= 0 && src < kLanes) {
                double g[D];
#pragma unroll
                for (int i = 0; i < D; ++i) g[i] = sv[k & 1][i][src];
                matvec_acc<D>(&sPw[which][k][0], g, s);
            }
        }
#pragma unroll
        for (int i = 0; i < D; ++i) sv[1][i][tid] = s[i];
        __syncthreads();
    };
    auto fetch = [&](const double* __restrict__ src, long long first, double (&el)[kQ][D]) {      // clamped: unconditional loads, all in flight
#pragma unroll
        for (int g = 0; g < kQ; ++g) {
            const long long idx = first + g < N ? first + g : N - 1;
#pragma unroll
            for (int i = 0; i < D; ++i) el[g][i] = src[idx * D + i];
        }
    };
    auto couple = [&](long long idx, const double (&mu)[D], double (&el)[D]) {                     // el -= C mu
        const double* __restrict__ C = cst + CL<D>::Bblk;                // (wave-uniform: scalar loads)
        const double* __restrict__ Cl = cst + CL<D>::Blb;
        const bool last = idx == N - 1;                                  // one lane: the ragged last workgroup
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double v = el[i];
#pragma unroll
            for (int l = 0; l < D; ++l) v = fma(-C[i * D + l], mu[l], v);
            if (__builtin_expect(last, 0)) {
                v = el[i];
#pragma unroll
                for (int l = 0; l < D; ++l) v = fma(-Cl[i * D + l], mu[l], v);
            }
            el[i] = v;
        }
    };
    const double* __restrict__ M = cst + CL<D>::pphi + kLogBlk * DD;
    const double* __restrict__ G = cst + CL<D>::pg + kLogBlk * DD;
    double carry[D], s[D], elb[kQ][D];
#pragma unroll
    for (int i = 0; i < D; ++i) carry[i] = MUb[i];
    if (POST && one) fetch(B0b, (long long)tid * kQ, elb);
    __syncthreads();
    // ---- forward
#pragma unroll 1
    for (long long sl = 0; sl < nslices; ++sl) {
        const long long base = sl * slice + (long long)tid * kQ;
        double el[kQ][D];
        fetch(Fb, base, el);
#pragma unroll
        for (int i = 0; i < D; ++i) s[i] = (tid == 0) ? carry[i] : 0.0;
#pragma unroll
        for (int g = 0; g < kQ; ++g)
            if (base + g < N) step(M, s, el[g]);
        block_scan(s, 0, true);
#pragma unroll
        for (int i = 0; i < D; ++i) {
            s[i] = (tid == 0) ? carry[i] : sv[1][i][tid > 0 ? tid - 1 : 0];
            carry[i] = sv[1][i][kLanes - 1];
        }
#pragma unroll
        for (int g = 0; g < kQ; ++g)
            if (base + g < N) {
                if (base + g > 0) {
#pragma unroll
                    for (int i = 0; i < D; ++i) MUb[(base + g) * D + i] = s[i];
                }
                if (POST && one) couple(base + g, s, elb[g]);
                step(M, s, el[g]);
            }
        __syncthreads();
    }
    if (!POST) return;
    if (!one) {
        __threadfence_block();
        __syncthreads();
    }
    // ---- backward
#pragma unroll
    for (int i = 0; i < D; ++i) carry[i] = lam_given ? LAMb[N * D + i] : 0.0;
    if (!lam_given && tid == 0) {
#pragma unroll
        for (int i = 0; i < D; ++i) LAMb[N * D + i] = 0.0;
    }
#pragma unroll 1
    for (long long sl = nslices - 1; sl >= 0; --sl) {
        const long long base = sl * slice + (long long)tid * kQ;
        if (!one) {
            double mu[kQ][D];
            fetch(MUb, base, mu);
            fetch(B0b, base, elb);
#pragma unroll
            for (int g = 0; g < kQ; ++g) couple(base + g, mu[g], elb[g]);
        }
        // the carry enters at the lane that owns the slice's LAST element (the last slice is ragged: the lanes behind it hold nothing, and
        // the scan's fixed powers would carry a non-zero lam -- a time shard's -- across them as if they held whole elements)
        const int inj = sl == nslices - 1 ? (int)((N - 1 - sl * slice) / kQ) : kLanes - 1;
#pragma unroll
        for (int i = 0; i < D; ++i) s[i] = (tid == inj) ? carry[i] : 0.0;
        const double* __restrict__ Glast = cst + CL<D>::GLb;      // (the series' last workgroup holds nvb <= 4096 steps)
#pragma unroll
        for (int g = kQ - 1; g >= 0; --g)
            if (base + g < N) step(base + g == N - 1 ? Glast : G, s, elb[g]);
        block_scan(s, 1, false);
#pragma unroll
        for (int i = 0; i < D; ++i) {
            s[i] = (tid == inj) ? carry[i] : sv[1][i][tid < kLanes - 1 ? tid + 1 : kLanes - 1];
            carry[i] = sv[1][i][0];
        }
#pragma unroll
        for (int g = kQ - 1; g >= 0; --g)
            if (base + g < N) {
                step(base + g == N - 1 ? Glast : G, s, elb[g]);
#pragma unroll
                for (int i = 0; i < D; ++i) LAMb[(base + g) * D + i] = s[i];
            }
        __syncthreads();
    }
}

// pass 2: the stationary tiles with their carries -> sum r^2, posterior marginals
template <int D, bool POST>
__global__ __launch_bounds__(kBlkThreads, (D <= 4 ? 4 : 2)) void k_apply(const long long* __restrict__ hdr, const double* __restrict__ cst, const double* __restrict__ y,
                                               const double* __restrict__ Rnew, int rnew_per_step, const double* __restrict__ F,
                                               const double* __restrict__ B0, const double* __restrict__ Pw, const double* __restrict__ Lw,
                                               const double* __restrict__ MUb, const double* __restrict__ LAMb,
                                               const double* __restrict__ t_vb, double* __restrict__ mean, double* __restrict__ var,
                                               double* __restrict__ SSQ, long long T, long long ntiles, Tab tb) {
    if (hdr[0] == 0) return;
    if (POST && blockIdx.x == 0) {      // the extra workgroup (dispatched first): the head's backward recursion, one wave, beside the stationary tiles
        if (threadIdx.x < 64) head_backward<D>(tb, y, Rnew, rnew_per_step, mean, var, T, threadIdx.x);
        return;
    }
    const long long wg = (long long)blockIdx.x - (POST ? 1 : 0);
    __shared__ double sacc[kBlk], sMu[kBlk + 1][D], sLam[kBlk][D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long tile0 = hdr[1] + wg * kBlk;
    if (tile0 >= ntiles) return;
    const long long tile = tile0 + wave;
    double acc = 0.0;
    const bool lastwg = wg == nblk_max_for(hdr[1], ntiles) - 1;
    if (lastwg) {                // the ragged last workgroup: its tiles' carries by the sequential chain (one thread)
        if (threadIdx.x == 0) {
            double mu_in[D], lam_in[D], lout[D];
#pragma unroll
            for (int i = 0; i < D; ++i) {
                mu_in[i] = MUb[wg * D + i];
                lam_in[i] = POST ? LAMb[(wg + 1) * D + i] : 0.0;
            }
            block_carries<D>(cst, [&](int w, int i) { return F[(tile0 + w) * D + i]; }, [&](int w, int i) { return POST ? B0[(tile0 + w) * D + i] : 0.0; },
                             tile0, ntiles, mu_in, lam_in, sMu, sLam, lout);
        }
        __syncthreads();
    }
    if (tile < ntiles) {
        double cin[D], lin[D];
        if (lastwg) {
#pragma unroll
            for (int i = 0; i < D; ++i) {
                cin[i] = (lane == 0) ? sMu[wave][i] : 0.0;
                lin[i] = (lane == 63) ? sLam[wave][i] : 0.0;
            }
        } else {
            // every tile of the workgroup is full: mu_w = Phi^(512 w) mu_b + Pw, lam_w = G^(512 (kBlk-1-w)) lam_b - K_w mu_b + Lw (wave-uniform)
            constexpr int DD = D * D;
            double mub[D], lamb[D], mw[D], lw[D];
#pragma unroll
            for (int i = 0; i < D; ++i) {
                mub[i] = MUb[wg * D + i];
                lamb[i] = POST ? LAMb[(wg + 1) * D + i] : 0.0;
                mw[i] = Pw[tile * D + i];
                lw[i] = POST ? Lw[tile * D + i] : 0.0;
            }
            const int wu = __builtin_amdgcn_readfirstlane(wave);
            matvec_acc<D>(cst + CL<D>::PT + wu * DD, mub, mw);
            if (POST) {
                matvec_acc<D>(cst + CL<D>::GT + (kBlk - 1 - wu) * DD, lamb, lw);
                const double* __restrict__ K = cst + CL<D>::KW + wu * DD;
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int l = 0; l < D; ++l) lw[i] = fma(-K[i * D + l], mub[l], lw[i]);
            }
#pragma unroll
            for (int i = 0; i < D; ++i) {
                cin[i] = (lane == 0) ? mw[i] : 0.0;
                lin[i] = (lane == 63) ? lw[i] : 0.0;
            }
        }
        Coef<D> cf;
        cf.load(cst);
        const long long t0 = tile * kTile + lane * kSub;
        double yv[kSub], r[kSub], fend[D];
        load8(y, t0, T, yv);
        const long long left = T - t0;
        const int nvalid = left >= kSub ? kSub : (left > 0 ? (int)left : 0);
        tile_forward<D>(cf, cst + CL<D>::pphi, yv, nvalid, lane, cin, r, fend);
#pragma unroll
        for (int j = 0; j < kSub; ++j) acc = fma(r[j], r[j], acc);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) acc += __shfl_down(acc, off);
        if (POST) {
            double bend[D], lst[D];
            tile_backward<D>(cf, cst + CL<D>::pg, r, lane, lin, bend, lst);
            const double rn0 = Rnew[0];
            const double vb = cst[SS<D>::vb];
            const long long n1 = hdr[3];
            // the outputs leave in pairs, as the backward walk produces them, through the wave's LDS rows (flush_tile)
            __shared__ __attribute__((aligned(16))) double sOutM[kBlk][kTileRow], sOutV[kBlk][kTileRow];
            v2d* rm = reinterpret_cast<v2d*>(sOutM[wave]);
            v2d* rv = reinterpret_cast<v2d*>(sOutV[wave]);
            const int wb = tile_slot(lane);
            double mhi = 0.0, vhi = 0.0;
#pragma unroll
            for (int j = kSub - 1; j >= 0; --j) {
                double m = fma(-cf.rS, r[j], yv[j]);
#pragma unroll
                for (int k = 0; k < D; ++k) m = fma(cf.h[k], lst[k], m);
                const long long back = T - 1 - (t0 + j);
                const double vbt = (back >= 0 && back < n1) ? t_vb[back] : vb;
                const double v = vbt + (rnew_per_step ? ((t0 + j < T) ? Rnew[t0 + j] : 0.0) : rn0);
                if (j & 1) {
                    mhi = m;
                    vhi = v;
                } else {
                    v2d wm, wv;
                    wm.x = m; wm.y = mhi;
                    wv.x = v; wv.y = vhi;
                    rm[wb + (j >> 1)] = wm;
                    rv[wb + (j >> 1)] = wv;
                }
                double nl[D];
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    double w = cf.c[i] * r[j];
#pragma unroll
                    for (int k = 0; k < D; ++k) w = fma(cf.G[i][k], lst[k], w);
                    nl[i] = w;
                }
#pragma unroll
                for (int i = 0; i < D; ++i) lst[i] = nl[i];
            }
            lds_sync();
            flush_tile(mean, tile * kTile, T, rm, lane);
            flush_tile(var, tile * kTile, T, rv, lane);
        }
    }
    if (lane == 0) sacc[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < kBlk; ++w) t += sacc[w];
        SSQ[wg] = t;
    }
}

// log marginal likelihood from the pieces (fixed summation order) and the status of the call
// (the head's backward recursion of a posterior call runs as the extra workgroup of k_apply)
template <int D>
__global__ __launch_bounds__(256) void k_final(Tab tb, long long T, long long ntiles, double* result, const double* __restrict__ y,
                                               const double* __restrict__ Rnew, int rnew_per_step, double* __restrict__ mean,
                                               double* __restrict__ var) {
    constexpr int NT = 256;
    __shared__ double sm[NT];
    const int tid = threadIdx.x;
    if (tb.hdr[0] == 0) {
        if (tid == 0) {
            result[6] = kStatusNotApplicable;
            result[7] = (double)tb.hdr[2];
            if (tb.hdr[4] != 0) result[2] = 1.0;
        }
        return;
    }
    const long long N = nblk_max_for(tb.hdr[1], ntiles);
    double acc = 0.0;
    for (long long i = tid; i < N; i += 8 * NT) {     // eight loads in flight per lane, fixed summation order
        double v[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) v[g] = (i + g * NT < N) ? tb.SSQ[i + g * NT] : 0.0;
#pragma unroll
        for (int g = 0; g < 8; ++g) acc += v[g];
    }
    sm[tid] = acc;
    __syncthreads();
    for (int off = NT / 2; off >= 1; off >>= 1) {
        if (tid < off) sm[tid] += sm[tid + off];
        __syncthreads();
    }
    if (tid == 0) {
        const double* ss = tb.ssc;
        const long long n0 = tb.hdr[2];
        const double quad = tb.misc[0] + ss[SS<D>::iS] * sm[0];
        const double logdet = ss[SS<D>::LS] + (double)(T - n0) * ss[SS<D>::logS];
        result[0] = -0.5 * ((double)T * kLog2Pi + logdet + quad);
        result[6] = kStatusRan;
        result[7] = (double)n0;
    }
}


// =================================================================================================================================
// adjoint (gradient) pass: d logpdf / d (model blocks) of the stationary tiles by ONE backward recursion (see tgp_steady.hpp)
// =================================================================================================================================
// With psi_{t+1} = d logpdf / d mu_{t+1} behind step t, mu_t and r_t of every step of the stationary tiles, the gradient needs
//   SA = sum psi_{t+1} mu_t'   Sa = sum psi_{t+1}   Sk = sum psi_{t+1} r_t   Srm = sum r_t mu_t   Sr = sum r_t   SSQ = sum r_t^2
// (GradRec<D>: the order of the record).  A lane accumulates its 8 steps, a wave / a workgroup reduce in a fixed order, k_final_grad
// sums the workgroups.  The head's steps and the reverse sweep through the covariance recursion are the host's (tgp_api.hip).
template <int D>
__global__ __launch_bounds__(kBlkThreads, (D <= 4 ? 2 : 1)) void k_apply_grad(const long long* __restrict__ hdr, const double* __restrict__ cst, const double* __restrict__ y,
                                                    const double* __restrict__ F, const double* __restrict__ B0, const double* __restrict__ Pw,
                                                    const double* __restrict__ Lw, const double* __restrict__ MUb, const double* __restrict__ LAMb,
                                                    double* __restrict__ GS, double* __restrict__ SSQ, long long T, long long ntiles) {
    if (hdr[0] == 0) return;
    constexpr int DD = D * D, NS = GradRec<D>::NS;
    __shared__ double sMu[kBlk + 1][D], sLam[kBlk][D], sred[kBlk][NS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long tile0 = hdr[1] + (long long)blockIdx.x * kBlk;
    if (tile0 >= ntiles) return;
    const long long tile = tile0 + wave;
    const bool lastwg = (long long)blockIdx.x == nblk_max_for(hdr[1], ntiles) - 1;
    if (lastwg) {
        if (threadIdx.x == 0) {
            double mu_in[D], lam_in[D], lout[D];
#pragma unroll
            for (int i = 0; i < D; ++i) {
                mu_in[i] = MUb[(long long)blockIdx.x * D + i];
                lam_in[i] = LAMb[((long long)blockIdx.x + 1) * D + i];
            }
            block_carries<D>(cst, [&](int w, int i) { return F[(tile0 + w) * D + i]; }, [&](int w, int i) { return B0[(tile0 + w) * D + i]; }, tile0,
                             ntiles, mu_in, lam_in, sMu, sLam, lout);
        }
        __syncthreads();
    }
    double acc[NS];
#pragma unroll
    for (int q = 0; q < NS; ++q) acc[q] = 0.0;
    if (tile < ntiles) {
        double cin[D], lin[D];
        if (lastwg) {
#pragma unroll
            for (int i = 0; i < D; ++i) {
                cin[i] = (lane == 0) ? sMu[wave][i] : 0.0;
                lin[i] = (lane == 63) ? sLam[wave][i] : 0.0;
            }
        } else {
            double mub[D], lamb[D], mw[D], lw[D];
#pragma unroll
            for (int i = 0; i < D; ++i) {
                mub[i] = MUb[(long long)blockIdx.x * D + i];
                lamb[i] = LAMb[((long long)blockIdx.x + 1) * D + i];
                mw[i] = Pw[tile * D + i];
                lw[i] = Lw[tile * D + i];
            }
            const int wu = __builtin_amdgcn_readfirstlane(wave);
            matvec_acc<D>(cst + CL<D>::PT + wu * DD, mub, mw);
            matvec_acc<D>(cst + CL<D>::GT + (kBlk - 1 - wu) * DD, lamb, lw);
            const double* __restrict__ K = cst + CL<D>::KW + wu * DD;
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int l = 0; l < D; ++l) lw[i] = fma(-K[i * D + l], mub[l], lw[i]);
#pragma unroll
            for (int i = 0; i < D; ++i) {
                cin[i] = (lane == 0) ? mw[i] : 0.0;
                lin[i] = (lane == 63) ? lw[i] : 0.0;
            }
        }
        Coef<D> cf;
        cf.load(cst);
        const long long t0 = tile * kTile + lane * kSub;
        double yv[kSub], r[kSub], fend[D], mus[kSub][D];
        load8(y, t0, T, yv);
        const long long left = T - t0;
        const int nvalid = left >= kSub ? kSub : (left > 0 ? (int)left : 0);
        tile_forward<D, true>(cf, cst + CL<D>::pphi, yv, nvalid, lane, cin, r, fend, mus);
        double bend[D], psi[D];
        tile_backward<D>(cf, cst + CL<D>::pg, r, lane, lin, bend, psi);      // psi: behind the lane's last step
#pragma unroll
        for (int j = kSub - 1; j >= 0; --j) {
            const double rr = r[j];
#pragma unroll
            for (int i = 0; i < D; ++i) {
#pragma unroll
                for (int k = 0; k < D; ++k) acc[GradRec<D>::SA + i * D + k] = fma(psi[i], mus[j][k], acc[GradRec<D>::SA + i * D + k]);
                acc[GradRec<D>::Sa + i] += psi[i];
                acc[GradRec<D>::Sk + i] = fma(psi[i], rr, acc[GradRec<D>::Sk + i]);
                acc[GradRec<D>::Srm + i] = fma(rr, mus[j][i], acc[GradRec<D>::Srm + i]);
            }
            acc[GradRec<D>::Sr] += rr;
            acc[GradRec<D>::SSQ] = fma(rr, rr, acc[GradRec<D>::SSQ]);
            double nl[D];
#pragma unroll
            for (int i = 0; i < D; ++i) {
                double v = cf.c[i] * rr;
#pragma unroll
                for (int k = 0; k < D; ++k) v = fma(cf.G[i][k], psi[k], v);
                nl[i] = v;
            }
#pragma unroll
            for (int i = 0; i < D; ++i) psi[i] = nl[i];
        }
    }
#pragma unroll
    for (int q = 0; q < NS; ++q) {
        double v = acc[q];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off);
        if (lane == 0) sred[wave][q] = v;
    }
    __syncthreads();
    if (threadIdx.x < NS) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < kBlk; ++w) t += sred[w][threadIdx.x];
        GS[(long long)threadIdx.x * gridDim.x + blockIdx.x] = t;      // [NS][workgroups]: k_final_grad reads each sum's row contiguously
        if (threadIdx.x == GradRec<D>::SSQ) SSQ[blockIdx.x] = t;      // (k_final's reduction of the value reads it there)
    }
}

// sums of the workgroups (fixed order: wave q mod 16 takes sum q, its lanes stride over the workgroups) + everything else the host's half
// of the gradient needs, in ONE record: psi and mu at the head's end, n0 / head tiles / T, and the model blocks as they are bound.
template <int D>
__global__ __launch_bounds__(256) void k_final_grad(Tab tb, ModelDev m, long long T, long long ntiles, const double* __restrict__ GS, long long nbs, double* __restrict__ rec) {
    constexpr int DD = D * D, NS = GradRec<D>::NS;
    const int tid = threadIdx.x;
    const bool ok = tb.hdr[0] != 0;
    const long long N = ok ? nblk_max_for(tb.hdr[1], ntiles) : 0;
    // workgroup q < NS sums row q of GS (fixed order: strided partial sums, then a tree); workgroup 0 also packs the rest of the record
    {
        __shared__ double sm[256];
        const int q = blockIdx.x;
        double v = 0.0;
        for (long long b = tid; b < N; b += 256) v += GS[q * nbs + b];
        sm[tid] = v;
        __syncthreads();
        for (int off = 128; off >= 1; off >>= 1) {
            if (tid < off) sm[tid] += sm[tid + off];
            __syncthreads();
        }
        if (tid == 0) rec[q] = sm[0];
    }
    if (blockIdx.x != 0) return;
    if (tid < D) {
        rec[GradRec<D>::psi + tid] = ok ? tb.LAMb[tid] : 0.0;
        rec[GradRec<D>::mu + tid] = ok ? tb.MUb[tid] : 0.0;
    }
    if (tid == 0) {
        rec[GradRec<D>::meta + 0] = (double)tb.hdr[2];      // n0
        rec[GradRec<D>::meta + 1] = (double)tb.hdr[1];      // head tiles
        rec[GradRec<D>::meta + 2] = (double)T;
        rec[GradRec<D>::meta + 3] = ok ? 1.0 : 0.0;
    }
    double* md = rec + GradRec<D>::model;
    if (tid < DD) {
        md[tid] = m.A[tid];
        md[DD + D + tid] = m.Q[tid];
    }
    if (tid < D) {
        md[DD + tid] = m.a[tid];
        md[2 * DD + D + tid] = m.H[tid];
    }
    if (tid == 0) {
        md[2 * DD + 2 * D] = m.hh[0];
        md[2 * DD + 2 * D + 1] = m.R[0];
    }
    if (tid < D + D * (D + 1) / 2) md[2 * DD + 2 * D + 2 + tid] = m.x0[tid];
}

// =================================================================================================================================
// time shards (tgp_multi.hip): the segment as ONE element of the two recursions, and the carries the exchange brings back
// =================================================================================================================================
// After pass 1 and a carry pass under the segment's provisional boundary (rank 0: the head's mu, zero lam; other ranks: zero both): the
// predicted mean behind the segment's last step (the last workgroup's tiles walked by one thread: a segment that hands on its end has
// whole tiles only), the lam in front of its stationary steps, and the segment-level matrices of the constant block.
template <int D>
__global__ __launch_bounds__(64) void k_shard_pack(Tab tb, long long ntiles, int post, double* __restrict__ slot) {
    constexpr int DD = D * D;
    const int tid = threadIdx.x;
    const bool ok = tb.hdr[0] != 0;
    if (tid == 0) slot[0] = ok ? 1.0 : 0.0;
    if (!ok) {
        for (int q = 1 + tid; q < ShardSlot<D>::size; q += 64) slot[q] = 0.0;
        return;
    }
    const double* __restrict__ cst = tb.cst;
    if (tid < DD) {
        slot[ShardSlot<D>::P + tid] = cst[CL<D>::PSeg + tid];
        slot[ShardSlot<D>::G + tid] = cst[CL<D>::GSeg + tid];
        slot[ShardSlot<D>::B + tid] = cst[CL<D>::BSeg + tid];
    }
    if (tid == 0) {
        const long long th = tb.hdr[1], N = nblk_max_for(th, ntiles);
        const long long tile0 = th + (N - 1) * kBlk;
        const double* __restrict__ M = cst + CL<D>::pphi + kLogTile * DD;
        double mu[D];
#pragma unroll
        for (int i = 0; i < D; ++i) mu[i] = tb.MUb[(N - 1) * D + i];
        for (long long tile = tile0; tile < ntiles; ++tile) {
            double n[D];
#pragma unroll
            for (int i = 0; i < D; ++i) n[i] = tb.F[tile * D + i];
            matvec_acc<D>(M, mu, n);
#pragma unroll
            for (int i = 0; i < D; ++i) mu[i] = n[i];
        }
#pragma unroll
        for (int i = 0; i < D; ++i) {
            slot[ShardSlot<D>::F + i] = mu[i];
            slot[ShardSlot<D>::B0 + i] = post ? tb.LAMb[i] : 0.0;
        }
    }
}

// From the gathered slots of all ranks: this segment's real boundary -- mu at its first stationary step (ranks > 0; rank 0 keeps its
// head's), lam behind its last step -- by walking the chain of segment elements (W <= 64: one thread).  If ANY rank reports that the
// engine does not apply to its segment, this rank stops as well (every later kernel returns at once, k_final reports it).
//   mu_in(1) = F_0,  mu_in(q+1) = Phi^L_q mu_in(q) + F_q;     lam_in(W-1) = 0,  lam_in(q-1) = G^L_q lam_in(q) + B0_q - B_L_q mu_in(q)
template <int D>
__global__ __launch_bounds__(64) void k_shard_fold(Tab tb, long long ntiles, const double* __restrict__ gathered, int world, int rank, int post) {
    constexpr int NS = ShardSlot<D>::size;
    __shared__ double smu[64][D];
    if (threadIdx.x != 0) return;
    bool ok = tb.hdr[0] != 0;
    for (int q = 0; q < world; ++q) ok = ok && gathered[(size_t)q * NS] != 0.0;
    if (!ok) {
        tb.hdr[0] = 0;
        return;
    }
    double mu[D];
#pragma unroll
    for (int i = 0; i < D; ++i) mu[i] = 0.0;
    for (int q = 0; q + 1 < world; ++q) {          // mu_in of rank q + 1
        const double* __restrict__ sl = gathered + (size_t)q * NS;
        double n[D];
#pragma unroll
        for (int i = 0; i < D; ++i) n[i] = sl[ShardSlot<D>::F + i];
        if (q > 0) matvec_acc<D>(sl + ShardSlot<D>::P, mu, n);      // (rank 0's F is the mean itself)
#pragma unroll
        for (int i = 0; i < D; ++i) mu[i] = smu[q + 1][i] = n[i];
    }
    const long long N = nblk_max_for(tb.hdr[1], ntiles);
    if (rank > 0) {
#pragma unroll
        for (int i = 0; i < D; ++i) tb.MUb[i] = smu[rank][i];
    }
    double lam[D];
#pragma unroll
    for (int i = 0; i < D; ++i) lam[i] = 0.0;
    if (post) {
        for (int q = world - 1; q > rank; --q) {      // lam_in of rank q - 1
            const double* __restrict__ sl = gathered + (size_t)q * NS;
            double n[D];
#pragma unroll
            for (int i = 0; i < D; ++i) {
                double v = sl[ShardSlot<D>::B0 + i];
#pragma unroll
                for (int l = 0; l < D; ++l) v = fma(-sl[ShardSlot<D>::B + i * D + l], smu[q][l], v);
                n[i] = v;
            }
            matvec_acc<D>(sl + ShardSlot<D>::G, lam, n);
#pragma unroll
            for (int i = 0; i < D; ++i) lam[i] = n[i];
        }
    }
#pragma unroll
    for (int i = 0; i < D; ++i) tb.LAMb[N * D + i] = lam[i];
}
}  // namespace

// =================================================================================================================================
// host side
// =================================================================================================================================
struct Engine {
    void* slab = nullptr;
    size_t cap = 0;
    int d = 0;
    long long ntiles = 0;
    Tab tb{};
};

static int cov_mode_bits() {      // TGP_STEADY_COV=seq: the sequential covariance iteration in k_setup_core (default: the time-parallel one, d <= 4)
    static const int bits = [] {
        const char* v = std::getenv("TGP_STEADY_COV");
        return (v && std::strcmp(v, "seq") == 0) ? 4 : 0;
    }();
    return bits;
}
Engine* create() { return new Engine(); }
void destroy(Engine* e) {
    if (!e) return;
    if (e->slab) (void)tgp_alloc::dev_free(e->slab);
    delete e;
}
bool supports(int d) { return d >= 1 && d <= kMaxD; }

namespace {

hipError_t ensure(Engine* e, int d, long long ntiles) {
    if (e->slab && e->d == d && e->ntiles >= ntiles) return hipSuccess;
    const size_t DD = (size_t)d * d;
    const size_t nhmax = (size_t)kHeadMaxTiles * kTile;
    size_t off = 0;
    auto take = [&](size_t ndoubles) {
        const size_t o = off;
        off += (ndoubles + 7) & ~(size_t)7;      // 64-byte granules
        return o;
    };
    // constant block, as CL<D> lays it out
    const size_t ss_size = 2 * DD + 5 * (size_t)d + 6, c_pphi = (ss_size + 7) & ~(size_t)7, c_pg = c_pphi + kPowN * DD, c_size = c_pg + kPowN * DD + 4 * DD + 3 * kBlk * DD + 4 * DD + (size_t)2 * kSub * d;
    const size_t o_hdr = take(8), o_cst = take(c_size);
    const size_t o_hkA = take(nhmax * d), o_hrS = take(nhmax), o_hiS = take(nhmax), o_hG = take(nhmax * DD), o_hc = take(nhmax * d),
                 o_hvb = take(nhmax), o_hr = take(nhmax);
    const size_t o_sPf = take((nhmax + 1) * DD), o_sPp = take((nhmax + 1) * DD), o_sL = take((nhmax + 1) * DD), o_sPs = take((nhmax + 1) * DD);
    const size_t o_tvb = take(kTailMax), o_tPs = take((size_t)kTailMax * DD);
    const size_t nt = (size_t)ntiles + 8, nb = (size_t)(ntiles + kBlk - 1) / kBlk + 2;
    const size_t o_F = take(nt * d), o_B0 = take(nt * d), o_Pw = take(nt * d), o_Lw = take(nt * d), o_Fb = take(nb * d), o_B0b = take(nb * d), o_MUb = take(nb * d), o_LAMb = take(nb * d),
                 o_SSQ = take(nb), o_misc = take(16), o_GS = take(nb * (DD + 3 * (size_t)d + 2)), o_grec = take(grad_record_size(d));
    const size_t bytes = off * sizeof(double);
    if (bytes > e->cap) {
        if (e->slab) (void)tgp_alloc::dev_free(e->slab);
        e->slab = nullptr;
        e->cap = 0;
        hipError_t rc = tgp_alloc::dev_malloc(&e->slab, bytes);
        if (rc != hipSuccess) return rc;
        e->cap = bytes;
    }
    double* b = static_cast<double*>(e->slab);
    Tab& tb = e->tb;
    tb.hdr = reinterpret_cast<long long*>(b + o_hdr);
    tb.cst = b + o_cst; tb.ssc = tb.cst; tb.pw_phi = tb.cst + c_pphi; tb.pw_g = tb.cst + c_pg;
    tb.h_kA = b + o_hkA; tb.h_rS = b + o_hrS; tb.h_iS = b + o_hiS; tb.h_G = b + o_hG; tb.h_c = b + o_hc; tb.h_vb = b + o_hvb; tb.h_r = b + o_hr;
    tb.s_Pf = b + o_sPf; tb.s_Pp = b + o_sPp; tb.s_L = b + o_sL; tb.s_Ps = b + o_sPs;
    tb.t_vb = b + o_tvb; tb.t_Ps = b + o_tPs;
    tb.F = b + o_F; tb.B0 = b + o_B0; tb.Pw = b + o_Pw; tb.Lw = b + o_Lw; tb.Fb = b + o_Fb; tb.B0b = b + o_B0b; tb.MUb = b + o_MUb; tb.LAMb = b + o_LAMb; tb.SSQ = b + o_SSQ;
    tb.misc = b + o_misc;
    tb.GS = b + o_GS;
    tb.grec = b + o_grec;
    e->d = d;
    e->ntiles = ntiles;
    return hipSuccess;
}

struct Scope {
    const Hooks& hk;
    Scope(const Hooks& h, const char* name) : hk(h) { if (hk.begin) hk.begin(hk.ctx, name); }
    ~Scope() { if (hk.end) hk.end(hk.ctx); }
};

template <int D>
int enqueue_d(Engine* e, hipStream_t st, const ModelDev& m, const CallDev& c, const Hooks& hk, const ShardDev* sh, int phase) {
    static_assert(CL<D>::pphi == ((2 * D * D + 5 * D + 6 + 7) & ~7), "ensure() mirrors CL<D>");
    static_assert(CL<D>::size == CL<D>::pphi + 2 * kPowN * D * D + 4 * D * D + 3 * kBlk * D * D + 4 * D * D + 2 * kSub * D, "ensure() mirrors CL<D>");
    static_assert(ShardSlot<D>::size == 1 + 2 * D + 3 * D * D, "shard_slot_size() mirrors ShardSlot<D>");
    const long long T = c.T;
    const long long ntiles = (T + kTile - 1) / kTile;
    const bool post = c.mean != nullptr || (sh && sh->post);
    const Tab tb = e->tb;
    // workgroups of the stationary tiles if the head is one tile (a shard that does not start the series has no head)
    const unsigned blocks = (unsigned)((ntiles - ((sh && !sh->first) ? 0 : 1) + kBlk - 1) / kBlk);
    const unsigned cblocks = blocks <= 512 ? 1u : (blocks / 512 < 32 ? blocks / 512 : 32u);      // k_carry: one workgroup per 512 elements, at most 32
    if (sh) {
        // ---- a time shard, in two halves around the exchange of the segments' elements (tgp_multi.hip)
        if (blocks == 0) return (int)hipErrorInvalidValue;      // (the caller sends series of one tile to the general path)
        const int flags = (sh->first ? 0 : 1) | (sh->last ? 0 : 2);
        if (phase == 0) {
            { Scope s(hk, "k_steady_setup"); hipLaunchKernelGGL(k_setup_core<D>, dim3(1), dim3(D <= kCovScanMaxD ? 128 : 64), 0, st, m, tb, c.y, T, 0, flags | cov_mode_bits(), post ? 1 : 0, c.result); }
            if (post) {
                { Scope s(hk, "k_steady_reduce<posterior>"); hipLaunchKernelGGL((k_reduce<D, true>), dim3(blocks + 1), dim3(kBlkThreads / 2), 0, st, tb.hdr, tb.cst, c.y, tb.F, tb.B0, tb.Fb, tb.B0b, T, ntiles, m, tb, 1); }
                { Scope s(hk, "k_steady_carry<segment>"); hipLaunchKernelGGL((k_carry<D, true>), dim3(cblocks), dim3(512), 0, st, tb.hdr, tb.cst, tb.Fb, tb.B0b, tb.MUb, tb.LAMb, ntiles, 0); }
            } else {
                { Scope s(hk, "k_steady_reduce<logpdf>"); hipLaunchKernelGGL((k_reduce<D, false>), dim3(blocks), dim3(kBlkThreads / 2), 0, st, tb.hdr, tb.cst, c.y, tb.F, tb.B0, tb.Fb, tb.B0b, T, ntiles, m, tb, 0); }
                { Scope s(hk, "k_steady_carry<segment>"); hipLaunchKernelGGL((k_carry<D, false>), dim3(cblocks), dim3(512), 0, st, tb.hdr, tb.cst, tb.Fb, tb.B0b, tb.MUb, tb.LAMb, ntiles, 0); }
            }
            { Scope s(hk, "k_steady_shard_pack"); hipLaunchKernelGGL(k_shard_pack<D>, dim3(1), dim3(64), 0, st, tb, ntiles, post ? 1 : 0, sh->slot); }
            return (int)hipGetLastError();
        }
        { Scope s(hk, "k_steady_shard_fold"); hipLaunchKernelGGL(k_shard_fold<D>, dim3(1), dim3(64), 0, st, tb, ntiles, sh->gathered, sh->world, sh->rank, post ? 1 : 0); }
        if (post) {
            { Scope s(hk, "k_steady_carry<posterior>"); hipLaunchKernelGGL((k_carry<D, true>), dim3(cblocks), dim3(512), 0, st, tb.hdr, tb.cst, tb.Fb, tb.B0b, tb.MUb, tb.LAMb, ntiles, 1); }
            { Scope s(hk, "k_steady_apply<posterior>"); hipLaunchKernelGGL((k_apply<D, true>), dim3(blocks + 1), dim3(kBlkThreads), 0, st, tb.hdr, tb.cst, c.y, c.Rnew, c.rnew_per_step, tb.F, tb.B0, tb.Pw, tb.Lw, tb.MUb, tb.LAMb, tb.t_vb, c.mean, c.var, tb.SSQ, T, ntiles, tb); }
            { Scope s(hk, "k_steady_final<posterior>"); hipLaunchKernelGGL(k_final<D>, dim3(1), dim3(256), 0, st, tb, T, ntiles, c.result, c.y, c.Rnew, c.rnew_per_step, c.mean, c.var); }
        } else {
            { Scope s(hk, "k_steady_carry<logpdf>"); hipLaunchKernelGGL((k_carry<D, false>), dim3(cblocks), dim3(512), 0, st, tb.hdr, tb.cst, tb.Fb, tb.B0b, tb.MUb, tb.LAMb, ntiles, 0); }
            { Scope s(hk, "k_steady_apply<logpdf>"); hipLaunchKernelGGL((k_apply<D, false>), dim3(blocks), dim3(kBlkThreads), 0, st, tb.hdr, tb.cst, c.y, (const double*)nullptr, 0, tb.F, tb.B0, tb.Pw, tb.Lw, tb.MUb, tb.LAMb, tb.t_vb, (double*)nullptr, (double*)nullptr, tb.SSQ, T, ntiles, tb); }
            { Scope s(hk, "k_steady_final<logpdf>"); hipLaunchKernelGGL(k_final<D>, dim3(1), dim3(256), 0, st, tb, T, ntiles, c.result, c.y, c.Rnew, 0, (double*)nullptr, (double*)nullptr); }
        }
        return (int)hipGetLastError();
    }
    {
        Scope s(hk, "k_steady_setup");
        hipLaunchKernelGGL(k_setup_core<D>, dim3(1), dim3(D <= kCovScanMaxD ? 128 : 64), 0, st, m, tb, c.y, T, c.grad ? 1 : 0, cov_mode_bits(), (post && !c.grad) ? 1 : 0, c.result);
    }
    if (blocks == 0) {      // a series of one tile: the engine does not apply (k_setup_core: nh + 2 > T)
        Scope s(hk, "k_steady_final");
        hipLaunchKernelGGL(k_final<D>, dim3(1), dim3(256), 0, st, tb, T, ntiles, c.result, c.y, c.Rnew, 0, (double*)nullptr, (double*)nullptr);
        if (c.grad) hipLaunchKernelGGL(k_final_grad<D>, dim3(GradRec<D>::NS), dim3(256), 0, st, tb, m, T, ntiles, tb.GS, 1LL, tb.grec);
        return (int)hipGetLastError();
    }
    if (c.grad) {
        static_assert(GradRec<D>::size == 3 * D * D + 8 * D + 8 + D * (D + 1) / 2, "grad_record_size() mirrors GradRec<D>");
        { Scope s(hk, "k_steady_reduce<adjoint>"); hipLaunchKernelGGL((k_reduce<D, true>), dim3(blocks + 1), dim3(kBlkThreads / 2), 0, st, tb.hdr, tb.cst, c.y, tb.F, tb.B0, tb.Fb, tb.B0b, T, ntiles, m, tb, 0); }
        { Scope s(hk, "k_steady_carry<adjoint>"); hipLaunchKernelGGL((k_carry<D, true>), dim3(cblocks), dim3(512), 0, st, tb.hdr, tb.cst, tb.Fb, tb.B0b, tb.MUb, tb.LAMb, ntiles, 0); }
        { Scope s(hk, "k_steady_apply<adjoint>"); hipLaunchKernelGGL(k_apply_grad<D>, dim3(blocks), dim3(kBlkThreads), 0, st, tb.hdr, tb.cst, c.y, tb.F, tb.B0, tb.Pw, tb.Lw, tb.MUb, tb.LAMb, tb.GS, tb.SSQ, T, ntiles); }
        { Scope s(hk, "k_steady_final<adjoint>"); hipLaunchKernelGGL(k_final_grad<D>, dim3(GradRec<D>::NS), dim3(256), 0, st, tb, m, T, ntiles, tb.GS, (long long)blocks, tb.grec); }
        // (the value of the call: the usual reduction -- misc[0], LS and logS do not depend on the (G, c) of the block)
        { Scope s(hk, "k_steady_final<logpdf>"); hipLaunchKernelGGL(k_final<D>, dim3(1), dim3(256), 0, st, tb, T, ntiles, c.result, c.y, c.Rnew, 0, (double*)nullptr, (double*)nullptr); }
        return (int)hipGetLastError();
    }
    if (post) {
        { Scope s(hk, "k_steady_reduce<posterior>"); hipLaunchKernelGGL((k_reduce<D, true>), dim3(blocks + 1), dim3(kBlkThreads / 2), 0, st, tb.hdr, tb.cst, c.y, tb.F, tb.B0, tb.Fb, tb.B0b, T, ntiles, m, tb, 1); }
        { Scope s(hk, "k_steady_carry<posterior>"); hipLaunchKernelGGL((k_carry<D, true>), dim3(cblocks), dim3(512), 0, st, tb.hdr, tb.cst, tb.Fb, tb.B0b, tb.MUb, tb.LAMb, ntiles, 0); }
        { Scope s(hk, "k_steady_apply<posterior>"); hipLaunchKernelGGL((k_apply<D, true>), dim3(blocks + 1), dim3(kBlkThreads), 0, st, tb.hdr, tb.cst, c.y, c.Rnew, c.rnew_per_step, tb.F, tb.B0, tb.Pw, tb.Lw, tb.MUb, tb.LAMb, tb.t_vb, c.mean, c.var, tb.SSQ, T, ntiles, tb); }
        { Scope s(hk, "k_steady_final<posterior>"); hipLaunchKernelGGL(k_final<D>, dim3(1), dim3(256), 0, st, tb, T, ntiles, c.result, c.y, c.Rnew, c.rnew_per_step, c.mean, c.var); }
    } else {
        { Scope s(hk, "k_steady_reduce<logpdf>"); hipLaunchKernelGGL((k_reduce<D, false>), dim3(blocks), dim3(kBlkThreads / 2), 0, st, tb.hdr, tb.cst, c.y, tb.F, tb.B0, tb.Fb, tb.B0b, T, ntiles, m, tb, 0); }
        { Scope s(hk, "k_steady_carry<logpdf>"); hipLaunchKernelGGL((k_carry<D, false>), dim3(cblocks), dim3(512), 0, st, tb.hdr, tb.cst, tb.Fb, tb.B0b, tb.MUb, tb.LAMb, ntiles, 0); }
        { Scope s(hk, "k_steady_apply<logpdf>"); hipLaunchKernelGGL((k_apply<D, false>), dim3(blocks), dim3(kBlkThreads), 0, st, tb.hdr, tb.cst, c.y, (const double*)nullptr, 0, tb.F, tb.B0, tb.Pw, tb.Lw, tb.MUb, tb.LAMb, tb.t_vb, (double*)nullptr, (double*)nullptr, tb.SSQ, T, ntiles, tb); }
        { Scope s(hk, "k_steady_final<logpdf>"); hipLaunchKernelGGL(k_final<D>, dim3(1), dim3(256), 0, st, tb, T, ntiles, c.result, c.y, c.Rnew, 0, (double*)nullptr, (double*)nullptr); }
    }
    return (int)hipGetLastError();
}

}  // namespace

static int enqueue_any(Engine* e, hipStream_t stream, const ModelDev& m, const CallDev& c, const Hooks& hk, std::string* err, const ShardDev* sh, int phase);
int enqueue(Engine* e, hipStream_t stream, const ModelDev& m, const CallDev& c, const Hooks& hk, std::string* err) {
    return enqueue_any(e, stream, m, c, hk, err, nullptr, 0);
}
int enqueue_shard(Engine* e, hipStream_t stream, const ModelDev& m, const CallDev& c, const ShardDev& sh, int phase, const Hooks& hk, std::string* err) {
    if (phase == 0 ? sh.slot == nullptr : (sh.gathered == nullptr || sh.world < 1 || sh.world > 64 || sh.rank < 0 || sh.rank >= sh.world)) {
        if (err) *err = "tgp_steady::enqueue_shard: bad exchange buffer / world / rank";
        return (int)hipErrorInvalidValue;
    }
    return enqueue_any(e, stream, m, c, hk, err, &sh, phase);
}
size_t shard_slot_size(int d) { return (size_t)(1 + 2 * d + 3 * d * d); }
static int enqueue_any(Engine* e, hipStream_t stream, const ModelDev& m, const CallDev& c, const Hooks& hk, std::string* err, const ShardDev* sh, int phase) {
    if (!e || !supports(m.d) || c.T <= 0 || !c.y || !c.result || (c.mean && (!c.var || !c.Rnew))) {
        if (err) *err = "tgp_steady::enqueue: bad argument";
        return (int)hipErrorInvalidValue;
    }
    const long long ntiles = (c.T + kTile - 1) / kTile;
    hipError_t rc = ensure(e, m.d, ntiles);
    if (rc != hipSuccess) {
        if (err) *err = std::string("tgp_steady: hipMalloc: ") + hipGetErrorString(rc);
        return (int)rc;
    }
    int r = 0;
    switch (m.d) {
        case 1: r = enqueue_d<1>(e, stream, m, c, hk, sh, phase); break;
        case 2: r = enqueue_d<2>(e, stream, m, c, hk, sh, phase); break;
        case 3: r = enqueue_d<3>(e, stream, m, c, hk, sh, phase); break;
        case 4: r = enqueue_d<4>(e, stream, m, c, hk, sh, phase); break;
        case 5: r = enqueue_d<5>(e, stream, m, c, hk, sh, phase); break;
        case 6: r = enqueue_d<6>(e, stream, m, c, hk, sh, phase); break;
        case 7: r = enqueue_d<7>(e, stream, m, c, hk, sh, phase); break;
        case 8: r = enqueue_d<8>(e, stream, m, c, hk, sh, phase); break;
        default: r = (int)hipErrorInvalidValue;
    }
    if (r != 0 && err) *err = std::string("tgp_steady: launch: ") + hipGetErrorString((hipError_t)r);
    return r;
}

size_t grad_record_size(int d) { return (size_t)(3 * d * d + 8 * d + 8 + d * (d + 1) / 2); }
const double* grad_record(const Engine* e) { return (e && e->slab) ? e->tb.grec : nullptr; }

int last_info(Engine* e, hipStream_t stream, int64_t out[4]) {
    if (!e || !e->slab) return (int)hipErrorInvalidValue;
    long long h[8];
    double misc[16];
    hipError_t rc = hipMemcpyAsync(h, e->tb.hdr, sizeof h, hipMemcpyDeviceToHost, stream);
    if (rc == hipSuccess) rc = hipMemcpyAsync(misc, e->tb.misc, sizeof misc, hipMemcpyDeviceToHost, stream);
    if (rc == hipSuccess) rc = hipStreamSynchronize(stream);
    if (rc != hipSuccess) return (int)rc;
    if (std::getenv("TGP_STEADY_DEBUG") != nullptr)      // phases of k_setup in microseconds (100 MHz wall clock)
        fprintf(stderr, "[tgp steady2] n0 %lld n1 %lld head tiles %lld applies %lld | core: filter cov + gains %.1f us, tables %.1f, powers %.1f, head forward %.1f | side: tail + head cov %.1f, variances %.1f (starts %.1f after core's start)\n",
                h[2], h[3], h[1], h[0], (misc[9] - misc[8]) * 0.01, (misc[10] - misc[9]) * 0.01, (misc[11] - misc[10]) * 0.01, (misc[12] - misc[11]) * 0.01,
                (misc[14] - misc[13]) * 0.01, (misc[15] - misc[14]) * 0.01, (misc[13] - misc[8]) * 0.01);
    out[0] = h[2];
    out[1] = h[3];
    out[2] = h[1];
    out[3] = h[0];
    return 0;
}

}  // namespace tgp_steady
