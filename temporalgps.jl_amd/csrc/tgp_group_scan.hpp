// Block scans over FILTER elements in the group layout (eight lanes per element, lane j = column j): the lane-per-element
// scan kernels hold three to five d x d-matrix elements per lane in a Kogge-Stone round and spill from d = 5 on (a d = 8
// element is 152 doubles; ~1.5 ms per launch whatever the element count). Here a lane carries 3 d + 2 doubles per element.
// Same interface as k_scan_reduce / k_scan_apply (SoA elements in HBM, 256 elements per block, identity-padded), so the
// level structure in tgp_api.hip is untouched.
//   general products X Y        the left factor is PUBLISHED to the group's LDS tile (each lane writes its column), then
//                               column j of X Y (or X' Y) is a lane-local sum over the tile
//   (I + C J)^-1                Gauss-Jordan with partial pivoting, one pivot column broadcast per step
//   symmetric results           averaged with their transpose through the tile (as symmetrize<D> does per lane)
// f_combine / f_apply below follow tgp_math_body.inc operation by operation.
#pragma once

namespace TGP_NS {

template <int D> struct GElem {
    double A[D], C[D], J[D], b, eta;     // lane j: column j of Abar, C, J; element j of b, eta
};
template <int D> struct GState {
    double P[D], m;
};

template <int D> struct GroupOps {
    static constexpr int G = GroupGeom<D>::G, V0 = GroupGeom<D>::V0, V1 = GroupGeom<D>::V1;
    int j;
    bool act;
    double* tile;    // [0, 64) matrix i + 8 col ; [64, 72) and [72, 80) vectors
    __device__ __forceinline__ void publish(const double* col) const {
        wave_sync();
        TGP_GUNROLL for (int i = 0; i < D; ++i) tile[i + G * j] = col[i];
        wave_sync();
    }
    // with X published: out = X y[:, j]
    __device__ __forceinline__ void left(const double* y, double* out) const {
        TGP_GUNROLL for (int i = 0; i < D; ++i) {
            double acc = 0.0;
            TGP_GUNROLL for (int k = 0; k < D; ++k) acc = fma(tile[i + G * k], y[k], acc);
            out[i] = acc;
        }
    }
    // out = X' y[:, j]
    __device__ __forceinline__ void left_t(const double* y, double* out) const {
        TGP_GUNROLL for (int i = 0; i < D; ++i) {
            double acc = 0.0;
            TGP_GUNROLL for (int k = 0; k < D; ++k) acc = fma(tile[k + G * i], y[k], acc);
            out[i] = acc;
        }
    }
    // row j of the published matrix
    __device__ __forceinline__ void row(double* out) const {
        TGP_GUNROLL for (int k = 0; k < D; ++k) out[k] = act ? tile[j + G * k] : 0.0;
    }
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
    // S <- (S + S') / 2 for a matrix given by columns
    __device__ __forceinline__ void symmetrize(double* Sc) const {
        publish(Sc);
        TGP_GUNROLL for (int i = 0; i < D; ++i) Sc[i] = act ? 0.5 * (Sc[i] + tile[j + G * i]) : 0.0;
    }
    // X = M^-1 (columns); M destroyed. Gauss-Jordan, partial pivoting by rows (every lane swaps the same two rows).
    __device__ __forceinline__ void inverse(double* Mc, double* Xc) const {
        TGP_GUNROLL for (int i = 0; i < D; ++i) Xc[i] = (i == j) ? 1.0 : 0.0;
        TGP_GUNROLL for (int k = 0; k < D; ++k) {
            int piv = k;
            double best = fabs(Mc[k]);
            TGP_GUNROLL for (int i = k + 1; i < D; ++i) {
                const double v = fabs(Mc[i]);
                const bool gt = v > best;
                best = gt ? v : best;
                piv = gt ? i : piv;
            }
            piv = __shfl(piv, k, G);                     // lane k owns column k: its choice of pivot row
            TGP_GUNROLL for (int i = k + 1; i < D; ++i) {
                const bool sw = (piv == i);
                const double t = Mc[k], u = Mc[i], t2 = Xc[k], u2 = Xc[i];
                Mc[k] = sw ? u : t;
                Mc[i] = sw ? t : u;
                Xc[k] = sw ? u2 : t2;
                Xc[i] = sw ? t2 : u2;
            }
            // column k (after the swap) to everybody
            wave_sync();
            if (j == k) { TGP_GUNROLL for (int i = 0; i < D; ++i) tile[V0 + i] = Mc[i]; }
            wave_sync();
            double colk[D];
            TGP_GUNROLL for (int i = 0; i < D; ++i) colk[i] = tile[V0 + i];
            const double inv = 1.0 / colk[k];
            const double rm = Mc[k] * inv, rx = Xc[k] * inv;
            TGP_GUNROLL for (int i = 0; i < D; ++i) {
                if (i != k) {
                    Mc[i] = fma(-colk[i], rm, Mc[i]);
                    Xc[i] = fma(-colk[i], rx, Xc[i]);
                }
            }
            Mc[k] = rm;
            Xc[k] = rx;
        }
    }

    // out = later(b) o earlier(a)     (f_combine_impl)
    __device__ __forceinline__ void combine(const GElem<D>& a, const GElem<D>& b, GElem<D>& o) const {
        double M[D], X[D], T1[D], T2[D], t[D], v1[D], v2[D];
        publish(a.C);                                        // M = C_a J_b + I
        left(b.J, M);
        TGP_GUNROLL for (int i = 0; i < D; ++i) M[i] += (i == j) ? 1.0 : 0.0;
        inverse(M, X);                                       // X = (I + C_a J_b)^-1
        publish(b.A);                                        // T1 = A_b X
        left(X, T1);
        gather2(b.eta, a.b, v1, v2);                         // v1 = eta_b, v2 = b_a
        double uj = a.b, vj = b.eta;                         // u = b_a + C_a eta_b ; v = eta_b - J_b b_a  (C, J symmetric: rows = columns)
        TGP_GUNROLL for (int k = 0; k < D; ++k) { uj = fma(a.C[k], v1[k], uj); vj = fma(-b.J[k], v2[k], vj); }
        gather2(uj, vj, v1, v2);                             // v1 = u, v2 = v
        double zj = 0.0;                                     // z = X' v
        TGP_GUNROLL for (int k = 0; k < D; ++k) zj = fma(X[k], v2[k], zj);
        publish(T1);                                         // w = T1 u ; nA = T1 A_a ; T2 = T1 C_a
        double wj = 0.0;
        row(t);
        TGP_GUNROLL for (int k = 0; k < D; ++k) wj = fma(t[k], v1[k], wj);
        left(a.A, o.A);
        left(a.C, T2);
        o.b = wj + b.b;
        gather(zj, v1);                                      // eta = A_a' z + eta_a
        double ne = 0.0;
        TGP_GUNROLL for (int k = 0; k < D; ++k) ne = fma(a.A[k], v1[k], ne);
        o.eta = ne + a.eta;
        publish(b.A);                                        // C = T2 A_b' + C_b : needs row j of A_b
        row(t);
        publish(T2);
        TGP_GUNROLL for (int i = 0; i < D; ++i) {
            double acc = 0.0;
            TGP_GUNROLL for (int k = 0; k < D; ++k) acc = fma(tile[i + G * k], t[k], acc);
            o.C[i] = acc + b.C[i];
        }
        symmetrize(o.C);
        publish(b.J);                                        // J = A_a' X' (J_b A_a) + J_a
        left(a.A, T1);                                       // T1 := J_b A_a
        publish(X);
        left_t(T1, T2);                                      // T2 := X' J_b A_a
        publish(a.A);
        left_t(T2, T1);                                      // T1 := A_a' X' J_b A_a
        TGP_GUNROLL for (int i = 0; i < D; ++i) o.J[i] = T1[i] + a.J[i];
        symmetrize(o.J);
    }

    // state after the run e, given the state before it   (f_apply_impl)
    __device__ __forceinline__ void apply(const GElem<D>& e, const GState<D>& in, GState<D>& out) const {
        double M[D], X[D], T1[D], T2[D], t[D], v1[D];
        publish(in.P);                                       // M = P J + I
        left(e.J, M);
        TGP_GUNROLL for (int i = 0; i < D; ++i) M[i] += (i == j) ? 1.0 : 0.0;
        inverse(M, X);
        publish(e.A);                                        // T1 = A X
        left(X, T1);
        gather(e.eta, v1);                                   // u = m + P eta
        double uj = in.m;
        TGP_GUNROLL for (int k = 0; k < D; ++k) uj = fma(in.P[k], v1[k], uj);
        gather(uj, v1);
        publish(T1);                                         // m' = T1 u + b ; T2 = T1 P
        row(t);
        double wj = 0.0;
        TGP_GUNROLL for (int k = 0; k < D; ++k) wj = fma(t[k], v1[k], wj);
        left(in.P, T2);
        out.m = wj + e.b;
        publish(e.A);                                        // P' = T2 A' + C
        row(t);
        publish(T2);
        TGP_GUNROLL for (int i = 0; i < D; ++i) {
            double acc = 0.0;
            TGP_GUNROLL for (int k = 0; k < D; ++k) acc = fma(tile[i + G * k], t[k], acc);
            out.P[i] = acc + e.C[i];
        }
        symmetrize(out.P);
    }
};

template <int D> __device__ __forceinline__ void gelem_identity(GElem<D>& e, int j) {
    TGP_GUNROLL for (int i = 0; i < D; ++i) { e.A[i] = (i == j) ? 1.0 : 0.0; e.C[i] = 0.0; e.J[i] = 0.0; }
    e.b = 0.0;
    e.eta = 0.0;
}
// SoA element idx of an array of n  <->  group layout
template <int D> __device__ __forceinline__ void gelem_load(GElem<D>& e, const double* __restrict__ E, int64_t n, int64_t idx, int j, bool act) {
    constexpr int DD = D * D, DS = Dim<D>::DS;
    gelem_identity<D>(e, j);
    if (!act) {
        TGP_GUNROLL for (int i = 0; i < D; ++i) e.A[i] = 0.0;
        return;
    }
    TGP_GUNROLL for (int i = 0; i < D; ++i) {
        const int lo = i < j ? i : j, hi = i < j ? j : i;
        e.A[i] = E[(int64_t)(i + j * D) * n + idx];
        e.C[i] = E[(int64_t)(DD + D + hi * (hi + 1) / 2 + lo) * n + idx];
        e.J[i] = E[(int64_t)(DD + 2 * D + DS + hi * (hi + 1) / 2 + lo) * n + idx];
    }
    e.b = E[(int64_t)(DD + j) * n + idx];
    e.eta = E[(int64_t)(DD + D + DS + j) * n + idx];
}
template <int D> __device__ __forceinline__ void gelem_store(const GElem<D>& e, double* __restrict__ E, int64_t n, int64_t idx, int j) {
    constexpr int DD = D * D, DS = Dim<D>::DS;
    TGP_GUNROLL for (int i = 0; i < D; ++i) {
        E[(int64_t)(i + j * D) * n + idx] = e.A[i];
        if (i <= j) {
            E[(int64_t)(DD + D + j * (j + 1) / 2 + i) * n + idx] = e.C[i];
            E[(int64_t)(DD + 2 * D + DS + j * (j + 1) / 2 + i) * n + idx] = e.J[i];
        }
    }
    E[(int64_t)(DD + j) * n + idx] = e.b;
    E[(int64_t)(DD + D + DS + j) * n + idx] = e.eta;
}
template <int D> __device__ __forceinline__ void gstate_load(GState<D>& s, const double* __restrict__ S, int64_t n, int64_t idx, int j, bool act) {
    s.m = 0.0;
    TGP_GUNROLL for (int i = 0; i < D; ++i) s.P[i] = 0.0;
    if (!act) return;
    s.m = S[(int64_t)j * n + idx];
    TGP_GUNROLL for (int i = 0; i < D; ++i) {
        const int lo = i < j ? i : j, hi = i < j ? j : i;
        s.P[i] = S[(int64_t)(D + hi * (hi + 1) / 2 + lo) * n + idx];
    }
}
template <int D> __device__ __forceinline__ void gstate_store(const GState<D>& s, double* __restrict__ S, int64_t n, int64_t idx, int j) {
    S[(int64_t)j * n + idx] = s.m;
    TGP_GUNROLL for (int i = 0; i < D; ++i)
        if (i <= j) S[(int64_t)(D + j * (j + 1) / 2 + i) * n + idx] = s.P[i];
}

// ---------------------------------------------------------------- monoid policies for the scan kernels
// kPerLane doubles per lane are staged in LDS ([group][value][lane]) for the cross-group steps.
template <int D> struct GFilterMO {
    using Elem = GElem<D>;
    using State = GState<D>;
    static constexpr int G = GroupGeom<D>::G;
    static constexpr int kPerLane = 3 * D + 2, kPerGroup = kPerLane * G;
    __device__ __forceinline__ static void identity(Elem& e, int j, bool act) {
        gelem_identity<D>(e, j);
        if (!act) { TGP_GUNROLL for (int i = 0; i < D; ++i) e.A[i] = 0.0; }
    }
    __device__ __forceinline__ static void load(Elem& e, const double* E, int64_t n, int64_t idx, int j, bool act) { gelem_load<D>(e, E, n, idx, j, act); }
    __device__ __forceinline__ static void store(const Elem& e, double* E, int64_t n, int64_t idx, int j) { gelem_store<D>(e, E, n, idx, j); }
    __device__ __forceinline__ static void combine(const GroupOps<D>& op, const Elem& a, const Elem& b, Elem& o) { op.combine(a, b, o); }
    __device__ __forceinline__ static void apply(const GroupOps<D>& op, const Elem& e, const State& s, State& o) { op.apply(e, s, o); }
    __device__ __forceinline__ static void put(double* st, int g, int j, const Elem& e) {
        double* p = st + g * kPerGroup + j;
        TGP_GUNROLL for (int i = 0; i < D; ++i) { p[(i) * G] = e.A[i]; p[(D + i) * G] = e.C[i]; p[(2 * D + i) * G] = e.J[i]; }
        p[(3 * D) * G] = e.b;
        p[(3 * D + 1) * G] = e.eta;
    }
    __device__ __forceinline__ static void get(const double* st, int g, int j, Elem& e) {
        const double* p = st + g * kPerGroup + j;
        TGP_GUNROLL for (int i = 0; i < D; ++i) { e.A[i] = p[(i) * G]; e.C[i] = p[(D + i) * G]; e.J[i] = p[(2 * D + i) * G]; }
        e.b = p[(3 * D) * G];
        e.eta = p[(3 * D + 1) * G];
    }
};

// affine monoid with covariance: x' = E x + g + N(0, L)  (a_combine_impl / a_apply_impl, COV = true)
template <int D> struct GAElem {
    double E[D], L[D], g;        // lane j: column j of E and L, element j of g
};
template <int D> struct GAffineMO {
    using Elem = GAElem<D>;
    using State = GState<D>;
    static constexpr int G = GroupGeom<D>::G;
    static constexpr int kPerLane = 2 * D + 1, kPerGroup = kPerLane * G;
    __device__ __forceinline__ static void identity(Elem& e, int j, bool act) {
        TGP_GUNROLL for (int i = 0; i < D; ++i) { e.E[i] = (act && i == j) ? 1.0 : 0.0; e.L[i] = 0.0; }
        e.g = 0.0;
    }
    __device__ __forceinline__ static void load(Elem& e, const double* __restrict__ E, int64_t n, int64_t idx, int j, bool act) {
        constexpr int DD = D * D;
        identity(e, j, false);
        if (!act) return;
        TGP_GUNROLL for (int i = 0; i < D; ++i) {
            const int lo = i < j ? i : j, hi = i < j ? j : i;
            e.E[i] = E[(int64_t)(i + j * D) * n + idx];
            e.L[i] = E[(int64_t)(DD + D + hi * (hi + 1) / 2 + lo) * n + idx];
        }
        e.g = E[(int64_t)(DD + j) * n + idx];
    }
    __device__ __forceinline__ static void store(const Elem& e, double* __restrict__ E, int64_t n, int64_t idx, int j) {
        constexpr int DD = D * D;
        TGP_GUNROLL for (int i = 0; i < D; ++i) {
            E[(int64_t)(i + j * D) * n + idx] = e.E[i];
            if (i <= j) E[(int64_t)(DD + D + j * (j + 1) / 2 + i) * n + idx] = e.L[i];
        }
        E[(int64_t)(DD + j) * n + idx] = e.g;
    }
    // out = later(b) o earlier(a):  E = E_b E_a ; g = E_b g_a + g_b ; L = E_b L_a E_b' + L_b
    __device__ __forceinline__ static void combine(const GroupOps<D>& op, const Elem& a, const Elem& b, Elem& o) {
        double t[D], v[D], T1[D];
        op.gather(a.g, v);
        op.publish(b.E);
        op.row(t);                                           // row j of E_b
        double gj = b.g;
        TGP_GUNROLL for (int k = 0; k < D; ++k) gj = fma(t[k], v[k], gj);
        op.left(a.E, o.E);
        op.left(a.L, T1);
        o.g = gj;
        op.publish(T1);
        TGP_GUNROLL for (int i = 0; i < D; ++i) {
            double acc = 0.0;
            TGP_GUNROLL for (int k = 0; k < D; ++k) acc = fma(op.tile[i + G * k], t[k], acc);
            o.L[i] = acc + b.L[i];
        }
        op.symmetrize(o.L);
    }
    // m' = E m + g ; P' = E P E' + L
    __device__ __forceinline__ static void apply(const GroupOps<D>& op, const Elem& e, const State& s, State& o) {
        double t[D], v[D], T1[D];
        op.gather(s.m, v);
        op.publish(e.E);
        op.row(t);
        double mj = e.g;
        TGP_GUNROLL for (int k = 0; k < D; ++k) mj = fma(t[k], v[k], mj);
        op.left(s.P, T1);
        o.m = mj;
        op.publish(T1);
        TGP_GUNROLL for (int i = 0; i < D; ++i) {
            double acc = 0.0;
            TGP_GUNROLL for (int k = 0; k < D; ++k) acc = fma(op.tile[i + G * k], t[k], acc);
            o.P[i] = acc + e.L[i];
        }
        op.symmetrize(o.P);
    }
    __device__ __forceinline__ static void put(double* st, int g, int j, const Elem& e) {
        double* p = st + g * kPerGroup + j;
        TGP_GUNROLL for (int i = 0; i < D; ++i) { p[(i) * G] = e.E[i]; p[(D + i) * G] = e.L[i]; }
        p[(2 * D) * G] = e.g;
    }
    __device__ __forceinline__ static void get(const double* st, int g, int j, Elem& e) {
        const double* p = st + g * kPerGroup + j;
        TGP_GUNROLL for (int i = 0; i < D; ++i) { e.E[i] = p[(i) * G]; e.L[i] = p[(D + i) * G]; }
        e.g = p[(2 * D) * G];
    }
};

// REDUCE: Ehi[b] = E[256 b] o ... o E[256 b + 255]. Each group folds its 256 / NGRP consecutive elements, then a tree over the groups.
template <int D, class MO>
__global__ __launch_bounds__(256) void k_group_scan_reduce(const double* __restrict__ Ein, int64_t n, double* __restrict__ Ehi, int64_t nhi) {
    constexpr int G = GroupGeom<D>::G, NGRP = GroupGeom<D>::NGRP, EPG = 256 / NGRP;    // elements per group
    __shared__ double tiles[NGRP * GroupGeom<D>::LD];
    __shared__ double stage[NGRP * MO::kPerGroup];
    const int tid = threadIdx.x, j = tid & (G - 1), g = tid / G;
    GroupOps<D> op{j, j < D, tiles + g * GroupGeom<D>::LD};
    const int64_t base = (int64_t)blockIdx.x * 256 + (int64_t)g * EPG;
    typename MO::Elem acc, e, t;
    MO::identity(acc, j, op.act);
    for (int q = 0; q < EPG; ++q) {
        const int64_t idx = base + q;
        if (idx < n) {                       // uniform inside the group
            MO::load(e, Ein, n, idx, j, op.act);
            if (q == 0) acc = e;
            else { MO::combine(op, acc, e, t); acc = t; }
        }
    }
    for (int off = 1; off < NGRP; off <<= 1) {
        __syncthreads();
        MO::put(stage, g, j, acc);
        __syncthreads();
        if ((g & (2 * off - 1)) == 0 && base + (int64_t)off * EPG < n) {      // the partner group holds at least one element
            MO::get(stage, g + off, j, e);
            MO::combine(op, acc, e, t);
            acc = t;
        }
    }
    if (g == 0 && op.act) MO::store(acc, Ehi, nhi, (int64_t)blockIdx.x, j);
}

// APPLY: S[i] = apply(E[256 b] o ... o E[i-1], carry[b]); fin (top level, one block) = state after every element.
// Group g folds its elements, the group totals are scanned (Kogge-Stone over LDS), the group's start state is the
// exclusive prefix applied to the block carry, and its states follow one apply at a time.
template <int D, class MO>
__global__ __launch_bounds__(256) void k_group_scan_apply(const double* __restrict__ Ein, int64_t n, const double* __restrict__ carry,
                                                          int64_t ncarry, double* __restrict__ S, double* __restrict__ fin) {
    constexpr int G = GroupGeom<D>::G, NGRP = GroupGeom<D>::NGRP, EPG = 256 / NGRP;
    __shared__ double tiles[NGRP * GroupGeom<D>::LD];
    __shared__ double stage[NGRP * MO::kPerGroup];
    const int tid = threadIdx.x, j = tid & (G - 1), g = tid / G;
    GroupOps<D> op{j, j < D, tiles + g * GroupGeom<D>::LD};
    const int64_t base = (int64_t)blockIdx.x * 256 + (int64_t)g * EPG;
    typename MO::Elem tot, e, t;
    MO::identity(tot, j, op.act);
    for (int q = 0; q < EPG; ++q) {
        const int64_t idx = base + q;
        if (idx < n) {
            MO::load(e, Ein, n, idx, j, op.act);
            if (q == 0) tot = e;
            else { MO::combine(op, tot, e, t); tot = t; }
        }
    }
    for (int off = 1; off < NGRP; off <<= 1) {               // inclusive scan of the group totals
        __syncthreads();                                     // everybody has finished reading the previous round
        MO::put(stage, g, j, tot);
        __syncthreads();
        if (g >= off) {
            MO::get(stage, g - off, j, e);
            MO::combine(op, e, tot, t);
            tot = t;
        }
    }
    __syncthreads();
    MO::put(stage, g, j, tot);
    __syncthreads();
    GState<D> s, s2;
    gstate_load<D>(s, carry, ncarry, (int64_t)blockIdx.x, j, op.act);
    if (g > 0) {
        MO::get(stage, g - 1, j, e);                         // exclusive prefix of this group
        MO::apply(op, e, s, s2);
        s = s2;
    }
    for (int q = 0; q < EPG; ++q) {
        const int64_t idx = base + q;
        if (idx < n) {
            if (op.act) gstate_store<D>(s, S, n, idx, j);
            MO::load(e, Ein, n, idx, j, op.act);
            MO::apply(op, e, s, s2);
            s = s2;
        }
    }
    // the state after the block's last element: held by the group that owns element n - 1 (top level: one block)
    if (fin != nullptr) {
        const int64_t last = n - 1 - (int64_t)blockIdx.x * 256;
        if (last >= 0 && last < 256 && (int)(last / EPG) == g && op.act) gstate_store<D>(s, fin, 1, 0, j);
    }
}

}  // namespace TGP_NS
