// Host half of the stationary-gain engine's one-launch path (round 4): everything about a call that does not depend on the observations.
//
// For an LTI model with one noise variance the covariance half of the reference recursion (predict lgc.jl:46-52, the scalar update
// lgc.jl:247-257, invert_dynamics lgssm.jl:231-238, the Reverse step_marginals lgssm.jl:111-115) never sees y.  Round 3 ran it in a
// one-wave kernel (36 us at d = 3, 222 us at d = 8: a dependent chain on a 2.4 GHz lane); here the host runs it in double, inside the
// call (nothing is kept between calls), in a few microseconds:
//   * the filtered covariance to its fixed point (n0 steps, the 2-ulp criterion of tgp_steady.hip) with the per-step gains of that head,
//   * the smoothed variances of the head and of the last n1 steps (the smoother's transient from the final filtered state),
//   * the stationary closed-loop matrices Phi = A - (A K) h' and G (smoother gain) in MODAL form: Phi = V M V^-1 with M block diagonal
//     (1 x 1 blocks for real eigenvalues, 2 x 2 rotation-scaling blocks for complex pairs), so that both mean recursions cost O(d) per
//     step instead of O(d^2) and every power M^n is an element-wise (complex) power.  The decomposition is accepted only if the
//     eigenvector matrices are well conditioned and reproduce Phi, G and the impulse response h' Phi^j (A K) to ~1e-13; otherwise
//     (near-defective closed loops, e.g. a sum of two IDENTICAL kernels) the call stays on the dense kernels of tgp_steady.hip.
//   * the halo length h: the number of steps after which Phi^h, G^h have decayed below 2^-64 -- a workgroup of the one-launch kernel
//     starts h steps early from a zero state and stops h steps late.
// Plain C++ (no HIP): compiled into libtgp_hip.so and exercised on the CPU tier through tgp_steady_plan_debug (tests/test_steady_plan.py).
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>

namespace tgp_plan {

constexpr int kMaxD = 8;
constexpr int kN0Max = 623;        // head steps with gains of their own the one-launch path accepts (nhs <= 640)
constexpr int kHeadMax = 640;      // nhs <= kHeadMax
constexpr int kTailMax = 2048;
constexpr int kHaloMax = 1536;     // longest halo (steps) served by the one-launch kernel
constexpr double kTol = 4.5e-16;   // "no longer changes": 2 ulp relative to the element's natural scale (as tgp_steady.hip)
constexpr double kCondMax = 1e5;
constexpr int kSub = 8;            // steps per lane of a tile

enum Why : int { kOk = 0, kNotSettled = 1, kNotPD = 2, kTooShort = 3, kIllConditioned = 4, kSlowMixing = 5, kTailLong = 6, kEigFail = 7 };

// Everything the one-launch kernel reads through its kernel arguments (uniform values: scalar loads).  Components are ordered complex
// pairs first -- positions (0, 1), (2, 3), ... -- then real modes; partner(i) = i ^ 1 where that exists, else i.
//   forward   z' = fd z + fo z_partner + fb u + fa,  u = y - hh,  r = u - fw . z          (z = V^-1 mu, mu the predicted mean)
//   backward  zeta' = gd zeta + go zeta_partner + gc r,  mean = y - rS r + gw . zeta      (zeta = W^-1 lam)
struct Modal {
    int d, n0, nhs, n1, halo, npair;
    double hh, rS, iS, logS, LS, vb;
    double fd[kMaxD], fo[kMaxD], fb[kMaxD], fa[kMaxD], fw[kMaxD];
    double gd[kMaxD], go[kMaxD], gc[kMaxD], gw[kMaxD];
    double fp8r[kMaxD], fp8i[kMaxD], gp8r[kMaxD], gp8i[kMaxD];              // M^8 (re, signed im), forward / backward
    double fp512r[kMaxD], fp512i[kMaxD], gp512r[kMaxD], gp512i[kMaxD];      // M^512
    double WJ[kSub][kMaxD];        // fw' M^j: the innovation j steps behind a lane's start state st is r0_j - WJ[j] . st
    double WG[kSub][kMaxD];        // gw' Mg^(7-j): the output of the lane's step j sees the lane's right-hand input through it
};

// What only the head wave of workgroup 0 and the last tiles read (pinned host memory, read by the device in place).
struct HeadTables {
    long long ready;               // 2 seq (+ 1: the tables half declined) once the tables of call `seq` are complete: written by the host AFTER the launch
    long long n1;                  // steps at the end of the series whose smoothed variance is still in its transient (tvb)
    double h[kMaxD];
    double mu0[kMaxD];             // V^-1 (A x0.m + a): the predicted mean of step 0 in the modal coordinates of the stationary closed loop
    double Wm[kMaxD * kMaxD];      // lam = Wm zeta (row-major)
    // per-step tables, t = 0..n0 (entry n0 is the stationary step), rows of d (d d) values packed by the model's d
    double kA[(kN0Max + 1) * kMaxD];      // V^-1 (A K_t - A K): what the head's forward recursion adds to the stationary one (zero at n0)
    double iS[kN0Max + 1], rS[kN0Max + 1];
    double G[(kN0Max + 1) * kMaxD * kMaxD], c[(kN0Max + 1) * kMaxD], vb[kN0Max + 1];
    double tvb[kTailMax];          // H Ps H' at step T-1-j, j < n1
};

struct Info {
    int why = 0, n0 = -1, n1 = -1, halo = 0;
    double cond_f = 0, cond_g = 0, rho = 0, resid = 0;
};

struct ModelHost {      // shared blocks, column-major as handed to tgp_model_set; x0P full d x d (upper triangle used)
    int d = 0;
    const double *A = nullptr, *a = nullptr, *Q = nullptr, *H = nullptr, *hh = nullptr, *R = nullptr, *x0m = nullptr, *x0P = nullptr;
};

namespace detail {

// (plain multiply-add: without -mfma std::fma is a libm call on x86-64 -- 40x the cost of the whole plan)
inline double pfma(double a, double b, double c) { return a * b + c; }

// (a bare complex type: std::complex multiplies and divides through __muldc3 / __divdc3 and takes moduli through hypot -- together
//  three quarters of the time of a 3 x 3 decomposition)
struct cplx {
    double re, im;
    cplx() : re(0.0), im(0.0) {}
    cplx(double r) : re(r), im(0.0) {}
    cplx(double r, double i) : re(r), im(i) {}
    double real() const { return re; }
    double imag() const { return im; }
    cplx& operator+=(const cplx& o) { re += o.re; im += o.im; return *this; }
    cplx& operator-=(const cplx& o) { re -= o.re; im -= o.im; return *this; }
    cplx& operator*=(const cplx& o) { const double r = re * o.re - im * o.im; im = re * o.im + im * o.re; re = r; return *this; }
    cplx& operator*=(double f) { re *= f; im *= f; return *this; }
    cplx& operator/=(double f) { const double g = 1.0 / f; re *= g; im *= g; return *this; }
};
inline cplx operator+(const cplx& a, const cplx& b) { return cplx(a.re + b.re, a.im + b.im); }
inline cplx operator-(const cplx& a, const cplx& b) { return cplx(a.re - b.re, a.im - b.im); }
inline cplx operator-(const cplx& a) { return cplx(-a.re, -a.im); }
inline cplx operator*(const cplx& a, const cplx& b) { return cplx(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
inline cplx operator*(double f, const cplx& a) { return cplx(f * a.re, f * a.im); }
inline cplx operator*(const cplx& a, double f) { return cplx(f * a.re, f * a.im); }
inline cplx operator/(const cplx& a, double f) { const double g = 1.0 / f; return cplx(a.re * g, a.im * g); }
inline cplx operator/(const cplx& a, const cplx& b) {
    const double g = 1.0 / (b.re * b.re + b.im * b.im);
    return cplx((a.re * b.re + a.im * b.im) * g, (a.im * b.re - a.re * b.im) * g);
}
inline cplx conj(const cplx& a) { return cplx(a.re, -a.im); }
inline double norm(const cplx& a) { return a.re * a.re + a.im * a.im; }
inline double cabs(const cplx& a) { return std::sqrt(a.re * a.re + a.im * a.im); }
inline cplx csqrt(const cplx& a) {      // principal square root
    const double m = cabs(a);
    if (m == 0.0) return cplx(0.0, 0.0);
    const double sr = std::sqrt(0.5 * (m + std::fabs(a.re)));
    const double si = 0.5 * a.im / sr;
    return (a.re >= 0.0) ? cplx(sr, si) : cplx(std::fabs(si), a.im >= 0.0 ? sr : -sr);
}

// Complex Schur form of a real n x n matrix (row-major in, n <= 8): Hessenberg reduction by Householder reflections, then single-shift QR
// with Wilkinson shifts and deflation.  T upper triangular, Q unitary, M = Q T Q^H.  Returns false if an eigenvalue does not converge.
inline bool complex_schur(int n, const double* M, cplx* T, cplx* Q) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            T[i * n + j] = M[i * n + j];
            Q[i * n + j] = (i == j) ? 1.0 : 0.0;
        }
    // Hessenberg
    for (int k = 0; k + 2 < n; ++k) {
        double alpha = 0.0;
        for (int i = k + 1; i < n; ++i) alpha += norm(T[i * n + k]);
        alpha = std::sqrt(alpha);
        if (alpha == 0.0) continue;
        cplx v[kMaxD];
        const cplx x0 = T[(k + 1) * n + k];
        const cplx ph = (cabs(x0) == 0.0) ? cplx(1.0) : x0 / cabs(x0);
        for (int i = 0; i < n; ++i) v[i] = (i > k) ? T[i * n + k] : cplx(0.0);
        v[k + 1] += ph * alpha;
        double vn = 0.0;
        for (int i = k + 1; i < n; ++i) vn += norm(v[i]);
        if (vn == 0.0) continue;
        // H = I - 2 v v^H / (v^H v): T <- H T H, Q <- Q H
        for (int j = 0; j < n; ++j) {
            cplx s = 0.0;
            for (int i = k + 1; i < n; ++i) s += conj(v[i]) * T[i * n + j];
            s *= 2.0 / vn;
            for (int i = k + 1; i < n; ++i) T[i * n + j] -= v[i] * s;
        }
        for (int i = 0; i < n; ++i) {
            cplx s = 0.0, sq = 0.0;
            for (int j = k + 1; j < n; ++j) {
                s += T[i * n + j] * v[j];
                sq += Q[i * n + j] * v[j];
            }
            s *= 2.0 / vn;
            sq *= 2.0 / vn;
            for (int j = k + 1; j < n; ++j) {
                T[i * n + j] -= s * conj(v[j]);
                Q[i * n + j] -= sq * conj(v[j]);
            }
        }
    }
    for (int i = 2; i < n; ++i)
        for (int j = 0; j + 1 < i; ++j) T[i * n + j] = 0.0;
    int hi = n - 1, iter = 0;
    const double eps = 2.220446049250313e-16;
    while (hi > 0) {
        // deflation: smallest l with a negligible subdiagonal below it
        int l = hi;
        while (l > 0) {
            const double s = cabs(T[(l - 1) * n + l - 1]) + cabs(T[l * n + l]);
            if (cabs(T[l * n + l - 1]) <= eps * (s == 0.0 ? 1.0 : s)) {
                T[l * n + l - 1] = 0.0;
                break;
            }
            --l;
        }
        if (l == hi) {
            --hi;
            iter = 0;
            continue;
        }
        if (++iter > 60) return false;
        // Wilkinson shift: eigenvalue of the trailing 2 x 2 closer to T[hi][hi]
        const cplx a = T[(hi - 1) * n + hi - 1], b = T[(hi - 1) * n + hi], c = T[hi * n + hi - 1], dd = T[hi * n + hi];
        const cplx tr = a + dd, det = a * dd - b * c;
        const cplx disc = csqrt(tr * tr - 4.0 * det);
        const cplx e1 = 0.5 * (tr + disc), e2 = 0.5 * (tr - disc);
        cplx sh = (cabs(e1 - dd) < cabs(e2 - dd)) ? e1 : e2;
        if (iter % 11 == 10) sh += cplx(cabs(c), cabs(c) * 0.5);      // exceptional shift
        // QR step on rows / columns l..hi by Givens rotations
        for (int i = l; i <= hi; ++i) T[i * n + i] -= sh;
        cplx cs[kMaxD], sn[kMaxD];
        for (int k = l; k < hi; ++k) {
            const cplx x = T[k * n + k], y = T[(k + 1) * n + k];
            const double r = std::sqrt(norm(x) + norm(y));
            cplx cc = 1.0, ss = 0.0;
            if (r != 0.0) {
                cc = x / r;
                ss = y / r;
            }
            cs[k] = cc;
            sn[k] = ss;
            for (int j = k; j < n; ++j) {      // rows k, k+1 <- G^H rows
                const cplx t1 = T[k * n + j], t2 = T[(k + 1) * n + j];
                T[k * n + j] = conj(cc) * t1 + conj(ss) * t2;
                T[(k + 1) * n + j] = -ss * t1 + cc * t2;
            }
        }
        for (int k = l; k < hi; ++k) {
            const cplx cc = cs[k], ss = sn[k];
            const int top = std::min(hi, k + 2);
            for (int i = 0; i <= top; ++i) {      // columns k, k+1 <- columns G
                const cplx t1 = T[i * n + k], t2 = T[i * n + k + 1];
                T[i * n + k] = t1 * cc + t2 * ss;
                T[i * n + k + 1] = -t1 * conj(ss) + t2 * conj(cc);
            }
            for (int i = 0; i < n; ++i) {
                const cplx t1 = Q[i * n + k], t2 = Q[i * n + k + 1];
                Q[i * n + k] = t1 * cc + t2 * ss;
                Q[i * n + k + 1] = -t1 * conj(ss) + t2 * conj(cc);
            }
        }
        for (int i = l; i <= hi; ++i) T[i * n + i] += sh;
    }
    return true;
}

// Real block-diagonal (modal) form of a real matrix: M = V B V^-1, B with 2 x 2 blocks [[re, im], [-im, re]] for complex pairs (first)
// and 1 x 1 blocks for real eigenvalues.  Outputs: re[i], im[i] (signed: +im for the first member of a pair, -im for the second, 0 for
// real modes), V and V^-1 (row-major), the number of pairs, the 1-norm condition number of V, the relative residual of M V = V B.
inline bool modal_form(int n, const double* M, double* re, double* im, double* V, double* Vinv, int* npair, double* cond, double* resid) {
    cplx T[kMaxD * kMaxD], Q[kMaxD * kMaxD];
    if (!complex_schur(n, M, T, Q)) return false;
    cplx lam[kMaxD], vec[kMaxD][kMaxD];
    double scale = 0.0;
    for (int i = 0; i < n * n; ++i) scale = std::max(scale, std::fabs(M[i]));
    if (!(scale > 0.0) || !std::isfinite(scale)) return false;
    for (int k = 0; k < n; ++k) {
        lam[k] = T[k * n + k];
        cplx x[kMaxD];
        for (int j = 0; j < n; ++j) x[j] = 0.0;
        x[k] = 1.0;
        for (int j = k - 1; j >= 0; --j) {
            cplx s = 0.0;
            for (int m = j + 1; m <= k; ++m) s += T[j * n + m] * x[m];
            cplx den = T[j * n + j] - lam[k];
            if (cabs(den) < 1e-300 + 2.3e-16 * scale) den = 2.3e-16 * scale;      // (a repeated eigenvalue: the conditioning check rejects the result)
            x[j] = -s / den;
        }
        double nn = 0.0;
        for (int i = 0; i < n; ++i) {
            cplx s = 0.0;
            for (int j = 0; j <= k; ++j) s += Q[i * n + j] * x[j];
            vec[k][i] = s;
            nn += norm(s);
        }
        nn = std::sqrt(nn);
        if (!(nn > 0.0) || !std::isfinite(nn)) return false;
        for (int i = 0; i < n; ++i) vec[k][i] /= nn;
    }
    // classify: complex pairs (im > 0 member keeps the vector) and real modes
    bool used[kMaxD];
    for (int k = 0; k < n; ++k) used[k] = false;
    int pos = 0, np = 0;
    double Vc[kMaxD][kMaxD];      // Vc[col][row]
    double rre[kMaxD], rim[kMaxD];
    for (int k = 0; k < n; ++k) {
        if (used[k]) continue;
        const double tol = 1e-10 * std::max(cabs(lam[k]), 1e-300);
        if (std::fabs(lam[k].imag()) <= tol) continue;
        // partner: the unused eigenvalue closest to the conjugate
        int best = -1;
        double bd = 1e300;
        for (int m = 0; m < n; ++m) {
            if (m == k || used[m]) continue;
            const double dist = cabs(lam[m] - conj(lam[k]));
            if (dist < bd) {
                bd = dist;
                best = m;
            }
        }
        if (best < 0 || bd > 1e-8 * cabs(lam[k])) return false;
        used[k] = used[best] = true;
        const int src = lam[k].imag() > 0.0 ? k : best;
        if (pos + 2 > n) return false;
        for (int i = 0; i < n; ++i) {
            Vc[pos][i] = vec[src][i].real();
            Vc[pos + 1][i] = vec[src][i].imag();
        }
        rre[pos] = rre[pos + 1] = lam[src].real();
        rim[pos] = lam[src].imag();
        rim[pos + 1] = -lam[src].imag();
        pos += 2;
        ++np;
    }
    for (int k = 0; k < n; ++k) {
        if (used[k]) continue;
        used[k] = true;
        // rotate the vector so that its largest component is real; the rest must then be real too
        int big = 0;
        for (int i = 1; i < n; ++i)
            if (cabs(vec[k][i]) > cabs(vec[k][big])) big = i;
        const cplx ph = conj(vec[k][big]) / cabs(vec[k][big]);
        double imax = 0.0;
        for (int i = 0; i < n; ++i) {
            const cplx w = vec[k][i] * ph;
            Vc[pos][i] = w.real();
            imax = std::max(imax, std::fabs(w.imag()));
        }
        if (imax > 1e-7) return false;
        rre[pos] = lam[k].real();
        rim[pos] = 0.0;
        ++pos;
    }
    if (pos != n) return false;
    // an odd number of components behind the pairs is fine; a pair must start at an even position (it does: pairs come first)
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) V[i * n + j] = Vc[j][i];
    // inverse by Gauss-Jordan with partial pivoting
    double Aug[kMaxD][2 * kMaxD];
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            Aug[i][j] = V[i * n + j];
            Aug[i][n + j] = (i == j) ? 1.0 : 0.0;
        }
    for (int c = 0; c < n; ++c) {
        int p = c;
        for (int i = c + 1; i < n; ++i)
            if (std::fabs(Aug[i][c]) > std::fabs(Aug[p][c])) p = i;
        if (!(std::fabs(Aug[p][c]) > 0.0)) return false;
        if (p != c)
            for (int j = 0; j < 2 * n; ++j) std::swap(Aug[p][j], Aug[c][j]);
        const double inv = 1.0 / Aug[c][c];
        for (int j = 0; j < 2 * n; ++j) Aug[c][j] *= inv;
        for (int i = 0; i < n; ++i) {
            if (i == c) continue;
            const double f = Aug[i][c];
            if (f == 0.0) continue;
            for (int j = 0; j < 2 * n; ++j) Aug[i][j] -= f * Aug[c][j];
        }
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) Vinv[i * n + j] = Aug[i][n + j];
    double n1 = 0.0, n1i = 0.0;
    for (int j = 0; j < n; ++j) {
        double s = 0.0, si = 0.0;
        for (int i = 0; i < n; ++i) {
            s += std::fabs(V[i * n + j]);
            si += std::fabs(Vinv[i * n + j]);
        }
        n1 = std::max(n1, s);
        n1i = std::max(n1i, si);
    }
    *cond = n1 * n1i;
    // residual of M V = V B
    double rmax = 0.0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double mv = 0.0;
            for (int k = 0; k < n; ++k) mv += M[i * n + k] * V[k * n + j];
            // (V B)_ij = V_ij re_j + V_i,partner * B[partner][j];  B[p][j] for a pair (0, 1): B = [[re, im], [-im, re]]
            const int pj = (j < 2 * np) ? (j ^ 1) : j;
            double vb = V[i * n + j] * rre[j];
            if (pj != j) vb += V[i * n + pj] * (-rim[j]);      // column j of B: B[pj][j] = -rim[j]  (j even: -im; j odd: +im)
            rmax = std::max(rmax, std::fabs(mv - vb));
        }
    *resid = rmax / scale;
    for (int i = 0; i < n; ++i) {
        re[i] = rre[i];
        im[i] = rim[i];
    }
    *npair = np;
    return true;
}

// reverse-time dynamics of one step (lgssm.jl:231-238): U'U = Symmetric(Pp) + 1e-10 I, G = (U \ (U' \ (A Pf)))', L = Pf - (U Gt)'(U Gt)
// C = A B, rows of C as sums of rows of B (the inner loop runs over a contiguous row: the host compiler vectorises it)
template <int D>
inline void mm(const double (&A)[D][D], const double (&B)[D][D], double (&C)[D][D]) {
    for (int i = 0; i < D; ++i) {
        double row[D];
        for (int j = 0; j < D; ++j) row[j] = 0.0;
        for (int k = 0; k < D; ++k) {
            const double a = A[i][k];
            for (int j = 0; j < D; ++j) row[j] += a * B[k][j];
        }
        for (int j = 0; j < D; ++j) C[i][j] = row[j];
    }
}
template <int D>
inline void transpose(const double (&A)[D][D], double (&At)[D][D]) {
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j) At[j][i] = A[i][j];
}
template <int D>
inline void mirror_upper(double (&P)[D][D]) {      // Symmetric(P): the upper triangle is the matrix
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < i; ++j) P[i][j] = P[j][i];
}

template <int D>
inline bool invert_dynamics(const double (&A)[D][D], const double (&Pf)[D][D], const double (&Pp)[D][D], double (&G)[D][D], double (&L)[D][D]) {
    // (every loop below runs over whole rows: U' U = Symmetric(Pp) + 1e-10 I by rows, the two triangular solves as row eliminations)
    double U[D][D], ru[D];      // (reciprocals of the pivots: a division costs as much as a 3 x 3 product here)
    bool ok = true;
    for (int i = 0; i < D; ++i) {
        double row[D];
        for (int j = 0; j < D; ++j) row[j] = (j >= i) ? Pp[i][j] : 0.0;
        row[i] += 1e-10;
        for (int k = 0; k < i; ++k) {
            const double f = U[k][i];
            for (int j = 0; j < D; ++j) row[j] -= f * U[k][j];      // (U[k][j] = 0 for j < k <= i: the lower part stays zero)
        }
        const double s = row[i];
        ok = ok && (s > 0.0);
        const double u = std::sqrt(s);
        ru[i] = 1.0 / u;
        for (int j = 0; j < D; ++j) U[i][j] = (j > i) ? row[j] * ru[i] : 0.0;
        U[i][i] = u;
    }
    // M = A Pf (full Pf, as the reference), X = U' \ M, Gt = U \ X
    double X[D][D];
    mm<D>(A, Pf, X);
    for (int i = 0; i < D; ++i) {
        for (int k = 0; k < i; ++k) {
            const double f = U[k][i];
            for (int c = 0; c < D; ++c) X[i][c] -= f * X[k][c];
        }
        for (int c = 0; c < D; ++c) X[i][c] *= ru[i];
    }
    for (int i = D - 1; i >= 0; --i) {
        for (int k = i + 1; k < D; ++k) {
            const double f = U[i][k];
            for (int c = 0; c < D; ++c) X[i][c] -= f * X[k][c];
        }
        for (int c = 0; c < D; ++c) X[i][c] *= ru[i];
    }
    double W[D][D];      // U Gt
    for (int i = 0; i < D; ++i) {
        double row[D];
        for (int c = 0; c < D; ++c) row[c] = 0.0;
        for (int k = i; k < D; ++k) {
            const double f = U[i][k];
            for (int c = 0; c < D; ++c) row[c] += f * X[k][c];
        }
        for (int c = 0; c < D; ++c) W[i][c] = row[c];
    }
    for (int i = 0; i < D; ++i) {      // L = Pf - W' W, G = Gt'
        double row[D];
        for (int j = 0; j < D; ++j) row[j] = 0.0;
        for (int k = 0; k < D; ++k) {
            const double f = W[k][i];
            for (int j = 0; j < D; ++j) row[j] += f * W[k][j];
        }
        for (int j = 0; j < D; ++j) {
            L[i][j] = Pf[i][j] - row[j];
            G[j][i] = X[i][j];
        }
    }
    return ok;
}

template <int D>
inline double quad_sym(const double (&h)[D], const double (&P)[D][D]) {      // h' Symmetric(P) h (upper triangle)
    double s = 0.0;
    for (int c = 0; c < D; ++c) {
        double v = 0.0;
        for (int r = 0; r < D; ++r) v = pfma(h[r], (r <= c ? P[r][c] : P[c][r]), v);
        s = pfma(v, h[c], s);
    }
    return s;
}

// P <- G Symmetric(P) G' + L
template <int D>
inline void smooth_cov_step(const double (&G)[D][D], const double (&L)[D][D], const double (&P)[D][D], double (&out)[D][D]) {
    double Ps[D][D], Gt[D][D], t1[D][D];
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j) Ps[i][j] = (i <= j) ? P[i][j] : P[j][i];
    transpose<D>(G, Gt);
    mm<D>(G, Ps, t1);
    mm<D>(t1, Gt, out);
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j) out[i][j] += L[i][j];
}

// element-wise power of the block-diagonal form: (re, im) -> (re, im)^n by repeated squaring; the sign convention of `im` is preserved
inline void modal_power(int d, const double* re, const double* im, long long n, double* pr, double* pi) {
    for (int i = 0; i < d; ++i) {
        cplx b(re[i], im[i]), acc(1.0, 0.0);
        long long e = n;
        while (e > 0) {
            if (e & 1) acc *= b;
            b *= b;
            e >>= 1;
        }
        pr[i] = acc.real();
        pi[i] = acc.imag();
    }
}

}  // namespace detail

// The plan of a call of T steps, in two halves.
//   build_core:   what the kernel needs before it can start -- the filtered covariance to its fixed point (tables kA, iS, rS as it goes), the
//                 stationary step's reverse-time dynamics, the stationary smoothed covariance (a Lyapunov equation, by doubling), the modal
//                 forms, the halo.  Info::why == kOk: the one-launch path applies as far as the core can tell.
//   build_tables: what only the head wave and the last tiles read -- the reverse-time dynamics of every head step, the smoothed variances of
//                 the head and of the last n1 steps, the head's gains in modal coordinates.  The caller may run it AFTER the launch (the
//                 kernel's head wave and last tiles wait for HeadTables::ready); it can still decline (a head step not positive definite,
//                 a smoother transient longer than kTailMax): returns a Why.
template <int D>
struct Work {
    double A[D][D], hv[D], R, kAss[D], Pss[D][D];
    double Pf[kN0Max + 2][D][D], Pp[kN0Max + 2][D][D];
    double Gss[D][D], Lss[D][D], Psinf[D][D];
    double Vi[kMaxD * kMaxD];
    int n0, nhs;
    unsigned long long stamp;      // of the build_core that filled this (unique in the process: whoever keeps a plan checks that the workspace is still its own)
};
template <int D>
inline Work<D>& work() {
    static thread_local Work<D> w;
    return w;
}
inline unsigned long long next_work_stamp() {
    static std::atomic<unsigned long long> c{0};
    return ++c;
}

template <int D>
inline Info build_core(const ModelHost& m, long long T, Modal& md, HeadTables& tab) {
    using namespace detail;
    Info info;
    Work<D>& wk = work<D>();
    wk.stamp = next_work_stamp();
    double A[D][D], hv[D];      // (locals: the compiler keeps them in registers; copies go to `wk` for build_tables)
    double Q[D][D], P[D][D], Pold2[D][D];
    for (int i = 0; i < D; ++i) {
        hv[i] = m.H[i];
        for (int k = 0; k < D; ++k) {
            A[i][k] = m.A[i + k * D];
            const int r = i < k ? i : k, c = i < k ? k : i;
            Q[i][k] = m.Q[r + c * D];            // Symmetric(Q), Symmetric(x0.P): upper triangles (as the device set-up)
            P[i][k] = m.x0P[r + c * D];
            Pold2[i][k] = 0.0;
        }
    }
    const double R = m.R[0];
    wk.R = R;
    std::memcpy(wk.A, A, sizeof A);
    std::memcpy(wk.hv, hv, sizeof hv);
    // ---- (a) filtered covariance to its fixed point; per-step (Pf, Pp) kept for the reverse-time dynamics
    int tc = -1, n0 = -1;
    double LS = 0.0, Sss = 1.0, kAss[D];
    bool bad = false;
    double At[D][D];
    transpose<D>(A, At);
    mirror_upper<D>(P);
    for (int t = 0; t <= kN0Max; ++t) {
        double t1[D][D], pp[D][D], V[D];
        mm<D>(A, P, t1);                  // A * Symmetric(P) (P is kept mirrored)
        mm<D>(t1, At, pp);
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < D; ++j) pp[i][j] += Q[i][j];
        double S = 0.0;
        for (int k = 0; k < D; ++k) {
            double v = 0.0;
            for (int l = 0; l < D; ++l) v = pfma(hv[l], pp[l][k], v);
            V[k] = v;
            S = pfma(v, hv[k], S);
        }
        S += R;
        if (!(S > 0.0)) {
            bad = true;
            break;
        }
        const double iS = 1.0 / S, rs = 1.0 / std::sqrt(S);
        for (int i = 0; i < D; ++i) {
            double v = 0.0;
            for (int k = 0; k < D; ++k) v = pfma(A[i][k], V[k] * iS, v);
            kAss[i] = v;
            tab.kA[t * D + i] = v;
        }
        tab.rS[t] = R * iS;
        tab.iS[t] = iS;
        std::memcpy(wk.Pf[t], P, sizeof P);
        std::memcpy(wk.Pp[t], pp, sizeof pp);
        Sss = S;
        if (tc >= 0) {      // the extra iteration from the settled covariance: the stationary step
            n0 = t;
            break;
        }
        LS += std::log(S);
        bool moved = false, cyc = t >= 1;
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < D; ++j) {
                const double Pn = pp[i][j] - (V[i] * rs) * (V[j] * rs);
                moved = moved || std::fabs(Pn - P[i][j]) > kTol * 0.5 * (pp[i][i] + pp[j][j]);
                cyc = cyc && (Pn == Pold2[i][j]);
                Pold2[i][j] = P[i][j];
                P[i][j] = Pn;
            }
        mirror_upper<D>(P);
        if (!moved || cyc) tc = t;
    }
    if (bad) {
        info.why = kNotPD;
        return info;
    }
    if (n0 < 0) {
        info.why = kNotSettled;
        return info;
    }
    info.n0 = n0;
    wk.n0 = n0;
    std::memcpy(wk.kAss, kAss, sizeof kAss);
    std::memcpy(wk.Pss, P, sizeof P);      // the settled filtered covariance
    const int nhs = 16 * ((n0 + 1 + 15) / 16);      // the head: [0, nhs), a whole number of 128-byte lines
    wk.nhs = nhs;
    if ((long long)nhs + 2 > T) {
        info.why = kTooShort;
        return info;
    }
    // ---- (b0) reverse-time dynamics of the stationary step
    if (!invert_dynamics<D>(A, wk.Pf[n0], wk.Pp[n0], wk.Gss, wk.Lss)) {
        info.why = kNotPD;
        return info;
    }
    double css[D];
    {
        double K[D], Sv = 0.0;
        for (int k = 0; k < D; ++k) {
            double v = 0.0;
            for (int l = 0; l < D; ++l) v = pfma(hv[l], wk.Pp[n0][l][k], v);
            K[k] = v;
            Sv = pfma(v, hv[k], Sv);
        }
        Sv += R;
        const double iSv = 1.0 / Sv;
        for (int r = 0; r < D; ++r) {
            double v = 0.0;
            for (int c = 0; c < D; ++c) v = pfma(wk.Gss[r][c], K[c] * iSv, v);
            css[r] = v;
        }
    }
    // ---- (c0) the stationary smoothed covariance: Ps = G Ps G' + L, i.e. sum_k G^k L G'^k, by doubling
    {
        double S[D][D], M[D][D];
        std::memcpy(S, wk.Lss, sizeof S);
        std::memcpy(M, wk.Gss, sizeof M);
        bool done = false;
        for (int it = 0; it < 48 && !done; ++it) {
            double t1[D][D], t2[D][D], M2[D][D];
            double mmax = 0.0;
            for (int i = 0; i < D; ++i)
                for (int j = 0; j < D; ++j) {
                    double v = 0.0, w = 0.0;
                    for (int k = 0; k < D; ++k) {
                        v = pfma(M[i][k], S[k][j], v);
                        w = pfma(M[i][k], M[k][j], w);
                    }
                    t1[i][j] = v;
                    M2[i][j] = w;
                    mmax = std::max(mmax, std::fabs(M[i][j]));
                }
            for (int i = 0; i < D; ++i)
                for (int j = 0; j < D; ++j) {
                    double v = 0.0;
                    for (int k = 0; k < D; ++k) v = pfma(t1[i][k], M[j][k], v);
                    t2[i][j] = v;
                }
            for (int i = 0; i < D; ++i)
                for (int j = i; j < D; ++j) {
                    const double v = S[i][j] + 0.5 * (t2[i][j] + t2[j][i]);
                    S[i][j] = S[j][i] = v;
                }
            std::memcpy(M, M2, sizeof M);
            done = mmax < 1e-10;      // (the term just added carried mmax^2)
        }
        if (!done) {
            info.why = kSlowMixing;
            return info;
        }
        std::memcpy(wk.Psinf, S, sizeof S);
    }
    const double vb_ss = quad_sym<D>(hv, wk.Psinf);
    // ---- (e) the stationary closed loop in modal form
    double Phi[D * D], Gs[D * D];
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j) {
            Phi[i * D + j] = A[i][j] - kAss[i] * hv[j];
            Gs[i * D + j] = wk.Gss[i][j];
        }
    double fre[kMaxD], fim[kMaxD], gre[kMaxD], gim[kMaxD], V[kMaxD * kMaxD], W[kMaxD * kMaxD], Wi[kMaxD * kMaxD];
    double(&Vi)[kMaxD * kMaxD] = wk.Vi;
    int npf = 0, npg = 0;
    double cf = 0, cg = 0, rf = 0, rg = 0;
    if (!modal_form(D, Phi, fre, fim, V, Vi, &npf, &cf, &rf) || !modal_form(D, Gs, gre, gim, W, Wi, &npg, &cg, &rg)) {
        info.why = kEigFail;
        return info;
    }
    info.cond_f = cf;
    info.cond_g = cg;
    info.resid = std::max(rf, rg);
    if (!(cf <= kCondMax) || !(cg <= kCondMax) || !(rf <= 1e-13 * cf) || !(rg <= 1e-13 * cg) || npf != npg) {
        info.why = kIllConditioned;
        return info;
    }
    double rho = 0.0;
    for (int i = 0; i < D; ++i) {
        rho = std::max(rho, std::sqrt(fre[i] * fre[i] + fim[i] * fim[i]));
        rho = std::max(rho, std::sqrt(gre[i] * gre[i] + gim[i] * gim[i]));
    }
    info.rho = rho;
    if (!(rho < 1.0)) {
        info.why = kSlowMixing;
        return info;
    }
    // halo: rho^h <= 2^-64
    int halo = 16;
    if (rho > 0.0) {
        const double need = std::ceil(-64.0 * std::log(2.0) / std::log(rho));
        if (!(need <= (double)kHaloMax)) {
            info.why = kSlowMixing;
            return info;
        }
        halo = 16 * (((int)need + 15) / 16);
        if (halo < 16) halo = 16;
    }
    info.halo = halo;
    // ---- (f) pack
    md.d = D;
    md.n0 = n0;
    md.nhs = nhs;
    md.n1 = -1;      // (build_tables)
    md.halo = halo;
    md.npair = npf;
    md.hh = m.hh[0];
    md.rS = R / Sss;
    md.iS = 1.0 / Sss;
    md.logS = std::log(Sss);
    md.LS = LS;
    md.vb = vb_ss;
    for (int i = 0; i < D; ++i) {
        md.fd[i] = fre[i];
        md.fo[i] = fim[i];
        md.gd[i] = gre[i];
        md.go[i] = gim[i];
        double b = 0.0, av = 0.0, w = 0.0, c = 0.0, o = 0.0;
        for (int k = 0; k < D; ++k) {
            b = pfma(Vi[i * D + k], kAss[k], b);
            av = pfma(Vi[i * D + k], m.a[k], av);
            w = pfma(hv[k], V[k * D + i], w);
            c = pfma(Wi[i * D + k], css[k], c);
            o = pfma(hv[k], W[k * D + i], o);
        }
        md.fb[i] = b;
        md.fa[i] = av;
        md.fw[i] = w;
        md.gc[i] = c;
        md.gw[i] = o;
    }
    modal_power(D, fre, fim, 8, md.fp8r, md.fp8i);
    modal_power(D, gre, gim, 8, md.gp8r, md.gp8i);
    modal_power(D, fre, fim, 512, md.fp512r, md.fp512i);
    modal_power(D, gre, gim, 512, md.gp512r, md.gp512i);
    // row vector times the block form: (x B)_j = x_j re_j + x_partner B[partner][j], B[partner][j] = -im_j
    auto row_times = [&](const double* re, const double* im, int np, const double* x, double* out) {
        for (int j = 0; j < D; ++j) {
            const int pj = (j < 2 * np) ? (j ^ 1) : j;
            double v = x[j] * re[j];
            if (pj != j) v = pfma(x[pj], -im[j], v);
            out[j] = v;
        }
    };
    {
        double x[kMaxD] = {0.0}, nx[kMaxD] = {0.0};
        for (int i = 0; i < D; ++i) x[i] = md.fw[i];
        for (int j = 0; j < kSub; ++j) {
            for (int i = 0; i < D; ++i) md.WJ[j][i] = x[i];
            row_times(fre, fim, npf, x, nx);
            for (int i = 0; i < D; ++i) x[i] = nx[i];
        }
        for (int i = 0; i < D; ++i) x[i] = md.gw[i];
        for (int j = kSub - 1; j >= 0; --j) {
            for (int i = 0; i < D; ++i) md.WG[j][i] = x[i];
            row_times(gre, gim, npg, x, nx);
            for (int i = 0; i < D; ++i) x[i] = nx[i];
        }
    }
    // impulse response check: h' Phi^j (A K) against fw' B^j fb, j < 32
    {
        double x[D], gmax = 0.0, emax = 0.0;
        for (int i = 0; i < D; ++i) x[i] = kAss[i];
        double zr[kMaxD] = {0.0};
        for (int i = 0; i < D; ++i) zr[i] = md.fb[i];
        for (int j = 0; j < 32; ++j) {
            double g = 0.0, gm = 0.0;
            for (int i = 0; i < D; ++i) {
                g = pfma(hv[i], x[i], g);
                gm = pfma(md.fw[i], zr[i], gm);
            }
            gmax = std::max(gmax, std::fabs(g));
            emax = std::max(emax, std::fabs(g - gm));
            double nx[D], nz[kMaxD];
            for (int i = 0; i < D; ++i) {
                double v = 0.0;
                for (int k = 0; k < D; ++k) v = pfma(Phi[i * D + k], x[k], v);
                nx[i] = v;
                const int pi_ = (i < 2 * npf) ? (i ^ 1) : i;
                nz[i] = pfma(fre[i], zr[i], (pi_ != i) ? fim[i] * zr[pi_] : 0.0);
            }
            for (int i = 0; i < D; ++i) {
                x[i] = nx[i];
                zr[i] = nz[i];
            }
        }
        if (!(emax <= 1e-12 * std::max(gmax, 1e-300))) {
            info.why = kIllConditioned;
            info.resid = emax / std::max(gmax, 1e-300);
            return info;
        }
    }
    // what the head needs of the modal forms: h, the first predicted mean in modal coordinates, W (lam = W zeta)
    {
        double m0[D];
        for (int i = 0; i < D; ++i) {
            tab.h[i] = hv[i];
            double v = m.a[i];
            for (int k = 0; k < D; ++k) {
                tab.Wm[i * D + k] = W[i * D + k];
                v = pfma(A[i][k], m.x0m[k], v);
            }
            m0[i] = v;
        }
        for (int i = 0; i < D; ++i) {
            double v = 0.0;
            for (int k = 0; k < D; ++k) v = pfma(Vi[i * D + k], m0[k], v);
            tab.mu0[i] = v;
        }
    }
    info.why = kOk;
    return info;
}

// The tables half in three stages, in the order the kernel's head wave wants them (the caller ships each as soon as it exists):
//   forward:  the head's gains in modal coordinates (cheap: one d x d product per step) -- the head's forward recursion can start;
//   backward: the reverse-time dynamics (G_t, c_t) of every head step (one d x d factorisation per step) -- its backward recursion;
//   variances: the smoothed variances of the head and of the last n1 steps -- only the writes of `var` at the two ends wait for them.
template <int D>
struct TablesWork {
    double sG[kN0Max + 2][D][D], sL[kN0Max + 2][D][D];
};
template <int D>
inline TablesWork<D>& tables_work() {
    static thread_local TablesWork<D> w;
    return w;
}

template <int D>
inline int build_tables_forward(Modal& md, HeadTables& tab) {
    using namespace detail;
    Work<D>& wk = work<D>();
    const int n0 = wk.n0;
    // the head's gains in the modal coordinates: z' = M z + fa + fb u + db_t r, db_t = V^-1 (A K_t - A K)
    for (int t = 0; t <= n0; ++t) {
        double dk[D], db[D];
        for (int k = 0; k < D; ++k) dk[k] = tab.kA[t * D + k] - wk.kAss[k];
        for (int i = 0; i < D; ++i) {
            double v = 0.0;
            for (int k = 0; k < D; ++k) v = pfma(wk.Vi[i * D + k], dk[k], v);
            db[i] = v;
        }
        for (int i = 0; i < D; ++i) tab.kA[t * D + i] = (t == n0) ? 0.0 : db[i];
    }
    (void)md;
    return kOk;
}

template <int D>
inline int build_tables_backward(Modal& md, HeadTables& tab) {
    using namespace detail;
    Work<D>& wk = work<D>();
    TablesWork<D>& tw = tables_work<D>();
    const int n0 = wk.n0;
    const double R = wk.R;
    double A[D][D], hv[D];
    std::memcpy(A, wk.A, sizeof A);
    std::memcpy(hv, wk.hv, sizeof hv);
    // ---- (b) reverse-time dynamics of the head steps (row n0: the stationary step's, from the core)
    std::memcpy(tw.sG[n0], wk.Gss, sizeof wk.Gss);
    std::memcpy(tw.sL[n0], wk.Lss, sizeof wk.Lss);
    for (int t = 0; t <= n0; ++t) {
        if (t < n0 && !invert_dynamics<D>(A, wk.Pf[t], wk.Pp[t], tw.sG[t], tw.sL[t])) return kNotPD;
        double K[D], Sv = 0.0;
        for (int k = 0; k < D; ++k) {
            double v = 0.0;
            for (int l = 0; l < D; ++l) v = pfma(hv[l], wk.Pp[t][l][k], v);
            K[k] = v;
            Sv = pfma(v, hv[k], Sv);
        }
        Sv += R;
        const double iSv = 1.0 / Sv;
        for (int r = 0; r < D; ++r) {
            double v = 0.0;
            for (int c = 0; c < D; ++c) {
                v = pfma(tw.sG[t][r][c], K[c] * iSv, v);
                tab.G[(size_t)t * D * D + r * D + c] = tw.sG[t][r][c];
            }
            tab.c[t * D + r] = v;
        }
    }
    (void)md;
    return kOk;
}

// The variances stage in its two halves: the TAIL (the smoothed variances of the last n1 steps: needs nothing of the backward stage -- the callers
// that have a kernel waiting for it, tgp_modal.hip `complete`, run it first) and the HEAD's (from the reverse-time dynamics of the backward stage).
template <int D>
inline int build_tables_tail(long long T, Modal& md, HeadTables& tab, Info& info) {
    using namespace detail;
    Work<D>& wk = work<D>();
    double hv[D], Gss[D][D], Lss[D][D];
    std::memcpy(hv, wk.hv, sizeof hv);
    std::memcpy(Gss, wk.Gss, sizeof Gss);
    std::memcpy(Lss, wk.Lss, sizeof Lss);
    // ---- (c) smoothed VARIANCES backwards from the final filtered state: Ps_j = sum_{k < j} G^k L G'^k + G^j P_ss G'^j, of which only
    //      h' Ps_j h is wanted -- with g_k = h' G^k (a row: O(d^2) per step instead of the O(d^3) of the matrix recursion)
    //      tvb_j = sum_{k < j} g_k L g_k' + g_j P_ss g_j'.  It has run into the stationary value once it no longer changes (2 ulp of it; the
    //      comparison with the Lyapunov solution only guards against a plateau far from the limit).
    int n1 = -1;
    {
        double g[D], Lsym[D][D], Psym[D][D];
        for (int i = 0; i < D; ++i) {
            g[i] = hv[i];
            for (int j = 0; j < D; ++j) {
                Lsym[i][j] = (i <= j) ? Lss[i][j] : Lss[j][i];
                Psym[i][j] = (i <= j) ? wk.Pss[i][j] : wk.Pss[j][i];
            }
        }
        auto quad = [&](const double (&M)[D][D], const double (&x)[D]) {
            double s = 0.0;
            for (int i = 0; i < D; ++i) {
                double v = 0.0;
                for (int j = 0; j < D; ++j) v += M[i][j] * x[j];
                s += x[i] * v;
            }
            return s;
        };
        double acc = 0.0, prev = 0.0, prev2 = 0.0;
        for (int jt = 0; jt < kTailMax; ++jt) {
            const double v = acc + quad(Psym, g);
            tab.tvb[jt] = v;
            if (jt >= 1 && (std::fabs(v - prev) <= kTol * std::fabs(v) || (jt >= 2 && v == prev2)) && std::fabs(v - md.vb) <= 1e-9 * std::fabs(md.vb)) {
                n1 = jt + 1;
                break;
            }
            prev2 = prev;
            prev = v;
            acc += quad(Lsym, g);
            double ng[D];
            for (int j = 0; j < D; ++j) ng[j] = 0.0;
            for (int i = 0; i < D; ++i) {
                const double f = g[i];
                for (int j = 0; j < D; ++j) ng[j] += f * Gss[i][j];
            }
            for (int j = 0; j < D; ++j) g[j] = ng[j];
        }
    }
    if (n1 < 0) return kTailLong;
    info.n1 = n1;
    md.n1 = n1;
    tab.n1 = n1;
    if ((long long)wk.nhs + n1 + 1 > T) return kTooShort;
    return kOk;
}
template <int D>
inline int build_tables_headvar(Modal& md, HeadTables& tab) {
    using namespace detail;
    Work<D>& wk = work<D>();
    TablesWork<D>& tw = tables_work<D>();
    const int n0 = wk.n0;
    double hv[D];
    std::memcpy(hv, wk.hv, sizeof hv);
    // ---- (d) smoothed variances of the head: Ps_{n0} = stationary, Ps_{t-1} = G_t Ps_t G_t' + L_t
    tab.vb[n0] = md.vb;
    {
        double Pc[D][D];
        std::memcpy(Pc, wk.Psinf, sizeof Pc);
        for (int t = n0; t >= 1; --t) {
            double pn[D][D];
            smooth_cov_step<D>(tw.sG[t], tw.sL[t], Pc, pn);
            std::memcpy(Pc, pn, sizeof pn);
            tab.vb[t - 1] = quad_sym<D>(hv, Pc);
        }
    }
    return kOk;
}
template <int D>
inline int build_tables_variances(long long T, Modal& md, HeadTables& tab, Info& info) {
    const int why = build_tables_tail<D>(T, md, tab, info);
    return why != kOk ? why : build_tables_headvar<D>(md, tab);
}

// ---- the head ON THE HOST (round 5): the kernel hands the head's observations over through pinned memory, the host runs the head's two
// recursions in the original coordinates with the per-step gains build_core left in `tab` (kA = A K_t, iS, rS -- build_tables_forward must NOT
// have run: it rewrites kA in modal coordinates) and the rows [G_t | c_t], vb of stages 1 and 2, and hands back the head's end state (modal:
// z0 = V^-1 mu_nhs) and the head's outputs.  What k_steady_one's head wave computed (head_forward / head_backward / head_variances), minus
// the PCIe pull of the tables and the one-wave dependent chains.
template <int D>
inline void modal_head_forward(const ModelHost& m, const Modal& md, const HeadTables& tab, const double* y, double* r_out, double* z0, double* quad) {
    using namespace detail;
    const Work<D>& wk = work<D>();
    double mu[D], nm[D];
    for (int i = 0; i < D; ++i) {
        double v = m.a[i];
        for (int k = 0; k < D; ++k) v = pfma(wk.A[i][k], m.x0m[k], v);
        mu[i] = v;
    }
    double q = 0.0;
    for (int t = 0; t < md.nhs; ++t) {
        const int ti = t < md.n0 ? t : md.n0;
        double r = y[t] - md.hh;
        for (int k = 0; k < D; ++k) r -= wk.hv[k] * mu[k];
        r_out[t] = r;
        q += r * r * tab.iS[ti];
        for (int i = 0; i < D; ++i) {
            double v = pfma(tab.kA[ti * D + i], r, m.a[i]);
            for (int k = 0; k < D; ++k) v = pfma(wk.A[i][k], mu[k], v);
            nm[i] = v;
        }
        for (int i = 0; i < D; ++i) mu[i] = nm[i];
    }
    for (int i = 0; i < D; ++i) {
        double v = 0.0;
        for (int k = 0; k < D; ++k) v = pfma(wk.Vi[i * D + k], mu[k], v);
        z0[i] = v;
    }
    *quad = q;
}
// zeta: the kernel's backward state entering the head from the right (modal); mean, vb [nhs]: y_t - (R / S_t) r_t + h' lam_t and h' Ps_t h
template <int D>
inline void modal_head_backward(const Modal& md, const HeadTables& tab, const double* y, const double* r, const double* zeta, double* mean, double* vb) {
    using namespace detail;
    double lam[D], nl[D];
    for (int i = 0; i < D; ++i) {
        double v = 0.0;
        for (int k = 0; k < D; ++k) v = pfma(tab.Wm[i * D + k], zeta[k], v);
        lam[i] = v;
    }
    for (int t = md.nhs - 1; t >= 0; --t) {
        const int ti = t < md.n0 ? t : md.n0;
        double o = y[t] - tab.rS[ti] * r[t];
        for (int k = 0; k < D; ++k) o = pfma(tab.h[k], lam[k], o);
        mean[t] = o;
        vb[t] = tab.vb[ti];
        const double* G = tab.G + (size_t)ti * D * D;
        for (int i = 0; i < D; ++i) {
            double v = tab.c[ti * D + i] * r[t];
            for (int k = 0; k < D; ++k) v = pfma(G[i * D + k], lam[k], v);
            nl[i] = v;
        }
        for (int i = 0; i < D; ++i) lam[i] = nl[i];
    }
}

// stage: 0 forward, 1 backward, 2 variances (in this order; each may decline)
template <int D>
inline int build_tables_stage(int stage, long long T, Modal& md, HeadTables& tab, Info& info) {
    if (stage == 0) return build_tables_forward<D>(md, tab);
    if (stage == 1) return build_tables_backward<D>(md, tab);
    return build_tables_variances<D>(T, md, tab, info);
}

template <int D>
inline int build_tables(long long T, Modal& md, HeadTables& tab, Info& info) {
    for (int stage = 0; stage < 3; ++stage) {
        const int why = build_tables_stage<D>(stage, T, md, tab, info);
        if (why != kOk) return why;
    }
    return kOk;
}

template <int D>
inline Info build(const ModelHost& m, long long T, Modal& md, HeadTables& tab) {
    Info info = build_core<D>(m, T, md, tab);
    if (info.why == kOk) info.why = build_tables<D>(T, md, tab, info);
    return info;
}

// ---- rand of an LTI model (lgssm.jl:65-91 with the draws supplied): x_t = A x_{t-1} + a + Lq eps_t, y_t = h' x_t + hh + sqrt(R) eta_t, a pure
// affine recursion.  Its matrix is the OPEN-loop transition -- for a Matern-3/2 or -5/2 block a Jordan block: no modal form -- so the plan
// keeps dense powers: A^(8 2^k) for the in-tile scan, A^512 and A^1024 for the tile chaining, the rows h' A^(j+1) that carry a lane's
// start state to its eight outputs, and the halo after which A^n has decayed below 2^-60 (found by multiplying, not from the spectral
// radius: ||A^n|| of a Jordan block carries a polynomial factor).  d <= kRandMaxD: everything is a kernel argument.
constexpr int kRandMaxD = 8;
struct RandPlan {
    int d = 0, halo = 0, why = kOk;
    double A[kRandMaxD * kRandMaxD], a[kRandMaxD], Lq[kRandMaxD * kRandMaxD], h[kRandMaxD], hh = 0.0, sR = 0.0;      // row-major; Lq lower
    double P[6][kRandMaxD * kRandMaxD], PT[2][kRandMaxD * kRandMaxD], WJ[kSub][kRandMaxD];
};

template <int D>
inline void build_rand(const ModelHost& m, RandPlan& rp) {
    using namespace detail;
    static_assert(D <= kRandMaxD, "");
    rp.d = D;
    rp.why = kOk;
    double A[D][D], Qs[D][D];
    for (int i = 0; i < D; ++i) {
        rp.a[i] = m.a[i];
        rp.h[i] = m.H[i];
        for (int k = 0; k < D; ++k) {
            A[i][k] = m.A[i + k * D];
            const int r = i < k ? i : k, c = i < k ? k : i;
            Qs[i][k] = m.Q[r + c * D] + (i == k ? 1e-9 : 0.0);      // Symmetric(Q + 1e-9 I) (lgc.jl:84-87)
        }
    }
    rp.hh = m.hh[0];
    if (!(m.R[0] >= 0.0)) {
        rp.why = kNotPD;
        return;
    }
    rp.sR = std::sqrt(m.R[0]);
    // lower Cholesky factor: cholesky(Symmetric(Q + 1e-9 I)).U'
    double Lc[D][D];
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j) Lc[i][j] = 0.0;
    for (int j = 0; j < D; ++j) {
        double v = Qs[j][j];
        for (int k = 0; k < j; ++k) v -= Lc[j][k] * Lc[j][k];
        if (!(v > 0.0)) {
            rp.why = kNotPD;
            return;
        }
        Lc[j][j] = std::sqrt(v);
        for (int i = j + 1; i < D; ++i) {
            double w = Qs[i][j];
            for (int k = 0; k < j; ++k) w -= Lc[i][k] * Lc[j][k];
            Lc[i][j] = w / Lc[j][j];
        }
    }
    auto put = [](const double (&M)[D][D], double* out) {
        for (int i = 0; i < D; ++i)
            for (int k = 0; k < D; ++k) out[i * D + k] = M[i][k];
    };
    put(A, rp.A);
    put(Lc, rp.Lq);
    // powers by squaring: A^8, then A^(8 2^k), A^512, A^1024
    double X[D][D], Y[D][D];
    std::memcpy(X, A, sizeof X);
    for (int q = 0; q < 3; ++q) {      // A^2, A^4, A^8
        mm<D>(X, X, Y);
        std::memcpy(X, Y, sizeof X);
    }
    double A16[D][D];
    for (int k = 0; k < 8; ++k) {
        if (k < 6) put(X, rp.P[k]);
        else put(X, rp.PT[k - 6]);
        if (k == 1) std::memcpy(A16, X, sizeof X);
        mm<D>(X, X, Y);
        std::memcpy(X, Y, sizeof X);
    }
    // the rows h' A^(j+1)
    {
        double x[D], nx[D];
        for (int i = 0; i < D; ++i) x[i] = m.H[i];
        for (int j = 0; j < kSub; ++j) {
            for (int k = 0; k < D; ++k) {
                double v = 0.0;
                for (int i = 0; i < D; ++i) v = pfma(x[i], A[i][k], v);
                nx[k] = v;
            }
            for (int k = 0; k < D; ++k) rp.WJ[j][k] = x[k] = nx[k];
        }
    }
    // halo: the smallest multiple of 16 with max |A^n| <= 2^-60 (relative to the states it multiplies: of the order of the stationary scale)
    {
        double M[D][D];
        std::memcpy(M, A16, sizeof M);
        int n = 16;
        const double tiny = std::ldexp(1.0, -60);
        for (;;) {
            double mx = 0.0;
            for (int i = 0; i < D; ++i)
                for (int k = 0; k < D; ++k) mx = std::max(mx, std::fabs(M[i][k]));
            if (!std::isfinite(mx)) {
                rp.why = kSlowMixing;
                return;
            }
            if (mx <= tiny) break;
            if (n >= kHaloMax) {
                rp.why = kSlowMixing;
                return;
            }
            mm<D>(M, A16, Y);
            std::memcpy(M, Y, sizeof M);
            n += 16;
        }
        rp.halo = n;
    }
}
inline void build_rand_any(const ModelHost& m, RandPlan& rp) {
    switch (m.d) {
        case 1: build_rand<1>(m, rp); return;
        case 2: build_rand<2>(m, rp); return;
        case 3: build_rand<3>(m, rp); return;
        case 4: build_rand<4>(m, rp); return;
        case 5: build_rand<5>(m, rp); return;
        case 6: build_rand<6>(m, rp); return;
        case 7: build_rand<7>(m, rp); return;
        case 8: build_rand<8>(m, rp); return;
    }
    rp.why = kEigFail;
}

// ---- _filter of an LTI model (lgssm.jl:171-187: the filtered means and covariances of every step, logpdf as a by-product).  The covariance
// half never sees the data: filtered covariances and gains run into their fixed point after n0 steps (the criterion of build_core); from
// there on the filtered mean is m_t = mu_t + K r_t with the predicted mean's stationary recursion mu' = Phi mu + a + (A K) u, Phi = A - (A K) h'
// -- a constant-coefficient affine recursion in u = y - hh, run by k_filter_one like rand's on DENSE powers of Phi.  The head (the first nhs
// steps, gains of their own) is run HERE, on the host, from the head's observations: a few microseconds beside a call that writes
// 8 (d + d^2) bytes per step.  d <= kRandMaxD (the powers are kernel arguments).
struct FilterPlan {
    int d = 0, n0 = -1, nhs = 0, halo = 0, why = kOk;
    double Phi[kRandMaxD * kRandMaxD], a[kRandMaxD], kA[kRandMaxD], K[kRandMaxD], h[kRandMaxD], hh = 0.0, iS = 0.0, logS = 0.0, LS = 0.0;
    double Pss[kRandMaxD * kRandMaxD];      // the settled filtered covariance (row-major; symmetric)
    double P[6][kRandMaxD * kRandMaxD], PT[2][kRandMaxD * kRandMaxD];      // Phi^(8 2^k); Phi^512, Phi^1024
};
template <int D>
struct FilterWork {      // the head's per-step tables, t = 0 .. n0
    double A[D][D], kA[kN0Max + 2][D], K[kN0Max + 2][D], iS[kN0Max + 2], Pf[kN0Max + 2][D][D];
};
template <int D>
inline FilterWork<D>& filter_work() {
    static thread_local FilterWork<D> w;
    return w;
}

template <int D>
inline void build_filter(const ModelHost& m, long long T, FilterPlan& fp) {
    using namespace detail;
    static_assert(D <= kRandMaxD, "");
    FilterWork<D>& fw = filter_work<D>();
    fp.d = D;
    fp.why = kOk;
    double A[D][D], At[D][D], Q[D][D], P[D][D], Pold2[D][D], hv[D];
    for (int i = 0; i < D; ++i) {
        hv[i] = m.H[i];
        fp.h[i] = m.H[i];
        fp.a[i] = m.a[i];
        for (int k = 0; k < D; ++k) {
            A[i][k] = m.A[i + k * D];
            const int r = i < k ? i : k, c = i < k ? k : i;
            Q[i][k] = m.Q[r + c * D];
            P[i][k] = m.x0P[r + c * D];
            Pold2[i][k] = 0.0;
        }
    }
    std::memcpy(fw.A, A, sizeof A);
    fp.hh = m.hh[0];
    const double R = m.R[0];
    transpose<D>(A, At);
    int tc = -1, n0 = -1;
    double LS = 0.0, Sss = 1.0;
    for (int t = 0; t <= kN0Max; ++t) {      // (the recursion and the criterion of build_core)
        double t1[D][D], pp[D][D], V[D];
        mm<D>(A, P, t1);
        mm<D>(t1, At, pp);
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < D; ++j) pp[i][j] += Q[i][j];
        double S = 0.0;
        for (int k = 0; k < D; ++k) {
            double v = 0.0;
            for (int l = 0; l < D; ++l) v = pfma(hv[l], pp[l][k], v);
            V[k] = v;
            S = pfma(v, hv[k], S);
        }
        S += R;
        if (!(S > 0.0)) {
            fp.why = kNotPD;
            return;
        }
        const double iS = 1.0 / S, rs = 1.0 / std::sqrt(S);
        for (int i = 0; i < D; ++i) {
            double v = 0.0;
            for (int k = 0; k < D; ++k) v = pfma(A[i][k], V[k] * iS, v);
            fw.kA[t][i] = v;
            fw.K[t][i] = V[i] * iS;
        }
        fw.iS[t] = iS;
        Sss = S;
        bool moved = false, cyc = t >= 1;
        double Pn[D][D];
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < D; ++j) {
                Pn[i][j] = pp[i][j] - (V[i] * rs) * (V[j] * rs);
                moved = moved || std::fabs(Pn[i][j] - P[i][j]) > kTol * 0.5 * (pp[i][i] + pp[j][j]);
                cyc = cyc && (Pn[i][j] == Pold2[i][j]);
            }
        std::memcpy(fw.Pf[t], Pn, sizeof Pn);      // the filtered covariance of step t
        if (tc >= 0) {      // the extra iteration from the settled covariance: the stationary step
            n0 = t;
            break;
        }
        LS += std::log(S);
        std::memcpy(Pold2, P, sizeof P);
        std::memcpy(P, Pn, sizeof P);
        mirror_upper<D>(P);
        if (!moved || cyc) tc = t;
    }
    if (n0 < 0) {
        fp.why = kNotSettled;
        return;
    }
    fp.n0 = n0;
    fp.nhs = 16 * ((n0 + 1 + 15) / 16);
    if ((long long)fp.nhs + 2 > T) {
        fp.why = kTooShort;
        return;
    }
    fp.iS = 1.0 / Sss;
    fp.logS = std::log(Sss);
    fp.LS = LS;
    double Phi[D][D];
    for (int i = 0; i < D; ++i) {
        fp.kA[i] = fw.kA[n0][i];
        fp.K[i] = fw.K[n0][i];
        for (int k = 0; k < D; ++k) {
            Phi[i][k] = A[i][k] - fw.kA[n0][i] * hv[k];
            fp.Phi[i * D + k] = Phi[i][k];
            fp.Pss[i * D + k] = 0.5 * (fw.Pf[n0][i][k] + fw.Pf[n0][k][i]);
        }
    }
    auto put = [](const double (&M)[D][D], double* out) {
        for (int i = 0; i < D; ++i)
            for (int k = 0; k < D; ++k) out[i * D + k] = M[i][k];
    };
    double X[D][D], Y[D][D], P16[D][D];
    std::memcpy(X, Phi, sizeof X);
    for (int q = 0; q < 3; ++q) {
        mm<D>(X, X, Y);
        std::memcpy(X, Y, sizeof X);
    }
    for (int k = 0; k < 8; ++k) {
        if (k < 6) put(X, fp.P[k]);
        else put(X, fp.PT[k - 6]);
        if (k == 1) std::memcpy(P16, X, sizeof X);
        mm<D>(X, X, Y);
        std::memcpy(X, Y, sizeof X);
    }
    {
        double M[D][D];
        std::memcpy(M, P16, sizeof M);
        int n = 16;
        const double tiny = std::ldexp(1.0, -60);
        for (;;) {
            double mx = 0.0;
            for (int i = 0; i < D; ++i)
                for (int k = 0; k < D; ++k) mx = std::max(mx, std::fabs(M[i][k]));
            if (!std::isfinite(mx) || n > kHaloMax) {
                fp.why = kSlowMixing;
                return;
            }
            if (mx <= tiny) break;
            mm<D>(M, P16, Y);
            std::memcpy(M, Y, sizeof M);
            n += 16;
        }
        fp.halo = n;
    }
}

// The head on the host: y [nhs] -> filtered means m [nhs][D], covariances Pc [nhs][D D], the predicted mean of step nhs, sum r^2 / S_t
template <int D>
inline void filter_head(const ModelHost& m, const FilterPlan& fp, const double* y, double* mout, double* Pout, double* mu_end, double* quad) {
    const FilterWork<D>& fw = filter_work<D>();
    double mu[D], nm[D];
    for (int i = 0; i < D; ++i) {
        double v = m.a[i];
        for (int k = 0; k < D; ++k) v += fw.A[i][k] * m.x0m[k];
        mu[i] = v;
    }
    double q = 0.0;
    for (int t = 0; t < fp.nhs; ++t) {
        const int ti = t < fp.n0 ? t : fp.n0;
        double r = y[t] - fp.hh;
        for (int k = 0; k < D; ++k) r -= fp.h[k] * mu[k];
        q += r * r * fw.iS[ti];
        for (int i = 0; i < D; ++i) {
            if (mout) mout[(size_t)t * D + i] = mu[i] + fw.K[ti][i] * r;
            double v = m.a[i] + fw.kA[ti][i] * r;
            for (int k = 0; k < D; ++k) v += fw.A[i][k] * mu[k];
            nm[i] = v;
        }
        if (Pout)
            for (int i = 0; i < D; ++i)
                for (int k = 0; k < D; ++k) Pout[(size_t)t * D * D + i * D + k] = 0.5 * (fw.Pf[ti][i][k] + fw.Pf[ti][k][i]);
        for (int i = 0; i < D; ++i) mu[i] = nm[i];
    }
    for (int i = 0; i < D; ++i) mu_end[i] = mu[i];
    *quad = q;
}
// posterior(model, y) of the head (lgssm.jl:193-221): the reverse-time transitions of the steps [0, nhs) -- G_t, L_t from
// invert_dynamics(x_(t-1), predict(x_(t-1))) (:231-238), g_t = m_(t-1) - G_t mu_t -- COLUMN-major blocks Gh, Lh [nhs][D D], gh [nhs + 1][D]
// (g of step nhs belongs to the settled transition but needs the head's last filtered mean), the settled Gss, Lss [D D], the predicted mean of
// step nhs and the head's sum r^2 / S_t.  false: a predicted covariance is not positive definite.
template <int D>
inline bool posterior_head(const ModelHost& m, const FilterPlan& fp, const double* y, double* Gh, double* gh, double* Lh, double* Gss, double* Lss,
                           double* mu_end, double* quad) {
    using namespace detail;
    const FilterWork<D>& fw = filter_work<D>();
    double A[D][D], At[D][D], Q[D][D], Pf[D][D], mprev[D], mu[D];
    for (int i = 0; i < D; ++i) {
        mprev[i] = m.x0m[i];
        for (int k = 0; k < D; ++k) {
            A[i][k] = fw.A[i][k];
            const int r = i < k ? i : k, c = i < k ? k : i;
            Q[i][k] = m.Q[r + c * D];
            Pf[i][k] = m.x0P[r + c * D];
        }
    }
    transpose<D>(A, At);
    auto put = [](const double (&M)[D][D], double* out) {      // column-major
        for (int i = 0; i < D; ++i)
            for (int k = 0; k < D; ++k) out[k * D + i] = M[i][k];
    };
    double q = 0.0, G[D][D], L[D][D];
    for (int t = 0; t <= fp.nhs; ++t) {
        double t1[D][D], Pp[D][D];
        mm<D>(A, Pf, t1);
        mm<D>(t1, At, Pp);
        for (int i = 0; i < D; ++i)
            for (int k = 0; k < D; ++k) Pp[i][k] += Q[i][k];
        if (!invert_dynamics<D>(A, Pf, Pp, G, L)) return false;
        for (int i = 0; i < D; ++i) {
            double v = m.a[i];
            for (int k = 0; k < D; ++k) v += A[i][k] * mprev[k];
            mu[i] = v;
        }
        for (int i = 0; i < D; ++i) {
            double v = mprev[i];
            for (int k = 0; k < D; ++k) v -= G[i][k] * mu[k];
            gh[(size_t)t * D + i] = v;
        }
        if (t == fp.nhs) break;
        put(G, Gh + (size_t)t * D * D);
        put(L, Lh + (size_t)t * D * D);
        const int ti = t < fp.n0 ? t : fp.n0;
        double r = y[t] - fp.hh;
        for (int k = 0; k < D; ++k) r -= fp.h[k] * mu[k];
        q += r * r * fw.iS[ti];
        for (int i = 0; i < D; ++i) {
            mprev[i] = mu[i] + fw.K[ti][i] * r;
            for (int k = 0; k < D; ++k) Pf[i][k] = 0.5 * (fw.Pf[ti][i][k] + fw.Pf[ti][k][i]);
        }
    }
    put(G, Gss);
    put(L, Lss);
    for (int i = 0; i < D; ++i) mu_end[i] = mu[i];
    *quad = q;
    return true;
}
inline bool posterior_head_any(const ModelHost& m, const FilterPlan& fp, const double* y, double* Gh, double* gh, double* Lh, double* Gss, double* Lss,
                               double* mu_end, double* quad) {
    switch (m.d) {
        case 1: return posterior_head<1>(m, fp, y, Gh, gh, Lh, Gss, Lss, mu_end, quad);
        case 2: return posterior_head<2>(m, fp, y, Gh, gh, Lh, Gss, Lss, mu_end, quad);
        case 3: return posterior_head<3>(m, fp, y, Gh, gh, Lh, Gss, Lss, mu_end, quad);
        case 4: return posterior_head<4>(m, fp, y, Gh, gh, Lh, Gss, Lss, mu_end, quad);
        case 5: return posterior_head<5>(m, fp, y, Gh, gh, Lh, Gss, Lss, mu_end, quad);
        case 6: return posterior_head<6>(m, fp, y, Gh, gh, Lh, Gss, Lss, mu_end, quad);
        case 7: return posterior_head<7>(m, fp, y, Gh, gh, Lh, Gss, Lss, mu_end, quad);
        case 8: return posterior_head<8>(m, fp, y, Gh, gh, Lh, Gss, Lss, mu_end, quad);
    }
    return false;
}
inline void build_filter_any(const ModelHost& m, long long T, FilterPlan& fp) {
    switch (m.d) {
        case 1: build_filter<1>(m, T, fp); return;
        case 2: build_filter<2>(m, T, fp); return;
        case 3: build_filter<3>(m, T, fp); return;
        case 4: build_filter<4>(m, T, fp); return;
        case 5: build_filter<5>(m, T, fp); return;
        case 6: build_filter<6>(m, T, fp); return;
        case 7: build_filter<7>(m, T, fp); return;
        case 8: build_filter<8>(m, T, fp); return;
    }
    fp.why = kEigFail;
}
inline void filter_head_any(const ModelHost& m, const FilterPlan& fp, const double* y, double* mout, double* Pout, double* mu_end, double* quad) {
    switch (m.d) {
        case 1: filter_head<1>(m, fp, y, mout, Pout, mu_end, quad); return;
        case 2: filter_head<2>(m, fp, y, mout, Pout, mu_end, quad); return;
        case 3: filter_head<3>(m, fp, y, mout, Pout, mu_end, quad); return;
        case 4: filter_head<4>(m, fp, y, mout, Pout, mu_end, quad); return;
        case 5: filter_head<5>(m, fp, y, mout, Pout, mu_end, quad); return;
        case 6: filter_head<6>(m, fp, y, mout, Pout, mu_end, quad); return;
        case 7: filter_head<7>(m, fp, y, mout, Pout, mu_end, quad); return;
        case 8: filter_head<8>(m, fp, y, mout, Pout, mu_end, quad); return;
    }
}

// ---- logpdf + posterior marginals of an LTI model on DENSE powers in both directions (round 5; k_smooth_one).  For the models
// build_core declines for want of a modal form (kIllConditioned: two summands with one length scale, ...; kEigFail).  Forwards the filter's
// plan above (mu' = Phi mu + a + (A K) u, r = u - h' mu); backwards the smoother in the innovations (lgssm.jl:111-115 re-associated as in
// build_core):  xi_t = c r_t + G xi_(t+1),  c = G K,  G the settled reverse-time transition of invert_dynamics (:231-238, jitter included),
// mean_t = y_t - (R / S) r_t + h' xi_(t+1) -- an affine recursion with ONE matrix, run by the kernel on the dense powers of G.  The head's two
// recursions (gains of their own) run on the host: forwards before the launch, backwards behind it from the xi the kernel hands back.
struct SmoothPlan {
    FilterPlan fp;
    int why = kOk, halo = 0, n1 = -1;
    double G[kRandMaxD * kRandMaxD], c[kRandMaxD], rS = 0.0, vb = 0.0;       // G row-major
    double GP[6][kRandMaxD * kRandMaxD], GPT[2][kRandMaxD * kRandMaxD];      // G^(8 2^k); G^512, G^1024
    double WJ[kSub][kRandMaxD];      // h' Phi^j: the innovation j steps behind a lane's start state st is r0_j - WJ[j] . st
    double WG[kSub][kRandMaxD];      // h' G^(7-j): the output of the lane's step j sees the lane's right-hand input through it
};
template <int D>
struct SmoothWork {
    double Gss[D][D], Lss[D][D], Psinf[D][D];
    double r[kHeadMax], G[kHeadMax + 1][D][D], vb[kHeadMax + 1];      // head: innovations, G_t (step t -> t - 1), h' Ps_t h
    double Uss[D][D], Ut[kHeadMax + 1][D][D];      // draws from the posterior: chol(L + 1e-9 I).U of the settled step and of every head step
    bool u_ok;
};
// U'U = Symmetric(M) + jitter I (the upper triangle of M is the matrix); U upper, zeros below.  false: not positive definite.
template <int D>
inline bool chol_upper(const double (&M)[D][D], double jitter, double (&U)[D][D]) {
    bool ok = true;
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j) U[i][j] = 0.0;
    for (int j = 0; j < D; ++j) {
        for (int i = 0; i <= j; ++i) {
            double acc = M[i][j] + (i == j ? jitter : 0.0);
            for (int k = 0; k < i; ++k) acc -= U[k][i] * U[k][j];
            if (i == j) {
                ok = ok && (acc > 0.0);
                U[j][j] = std::sqrt(acc > 0.0 ? acc : 1.0);
            } else {
                U[i][j] = acc / U[i][i];
            }
        }
    }
    return ok;
}
template <int D>
inline SmoothWork<D>& smooth_work() {
    static thread_local SmoothWork<D> w;
    return w;
}

// tvb [kTailMax]: h' Ps h of the steps T - 1 - j, j < n1 (the smoothed covariance's transient at the series' end)
template <int D>
inline void build_smooth(const ModelHost& m, long long T, SmoothPlan& sp, double* tvb, bool post = true) {
    using namespace detail;
    build_filter<D>(m, T, sp.fp);
    sp.why = sp.fp.why;
    if (sp.why != kOk) return;
    const FilterPlan& fp = sp.fp;
    if (!post) {      // logpdf only: the forward half and the rows h' Phi^j
        sp.halo = fp.halo;
        sp.n1 = 0;
        double x[D], nx[D];
        for (int i = 0; i < D; ++i) x[i] = fp.h[i];
        for (int j = 0; j < kSub; ++j) {
            for (int i = 0; i < D; ++i) sp.WJ[j][i] = x[i];
            for (int k = 0; k < D; ++k) {
                double v = 0.0;
                for (int i = 0; i < D; ++i) v = pfma(x[i], fp.Phi[i * D + k], v);
                nx[k] = v;
            }
            for (int i = 0; i < D; ++i) x[i] = nx[i];
        }
        return;
    }
    const FilterWork<D>& fw = filter_work<D>();
    SmoothWork<D>& sw = smooth_work<D>();
    const int n0 = fp.n0;
    double A[D][D], At[D][D], Q[D][D], Pf[D][D], Pp[D][D], t1[D][D], hv[D];
    for (int i = 0; i < D; ++i) {
        hv[i] = fp.h[i];
        for (int k = 0; k < D; ++k) {
            A[i][k] = fw.A[i][k];
            const int r = i < k ? i : k, c = i < k ? k : i;
            Q[i][k] = m.Q[r + c * D];
            Pf[i][k] = 0.5 * (fw.Pf[n0][i][k] + fw.Pf[n0][k][i]);
        }
    }
    transpose<D>(A, At);
    mm<D>(A, Pf, t1);
    mm<D>(t1, At, Pp);
    for (int i = 0; i < D; ++i)
        for (int k = 0; k < D; ++k) Pp[i][k] += Q[i][k];
    if (!invert_dynamics<D>(A, Pf, Pp, sw.Gss, sw.Lss)) {
        sp.why = kNotPD;
        return;
    }
    sp.rS = m.R[0] * fp.iS;
    for (int i = 0; i < D; ++i) {
        double v = 0.0;
        for (int k = 0; k < D; ++k) {
            v = pfma(sw.Gss[i][k], fp.K[k], v);
            sp.G[i * D + k] = sw.Gss[i][k];
        }
        sp.c[i] = v;
    }
    sw.u_ok = chol_upper<D>(sw.Lss, 1e-9, sw.Uss);      // (conditional_rand, lgc.jl:84-87: the noise of a reverse-time step of a posterior draw)
    // the stationary smoothed covariance: Ps = G Ps G' + L by doubling (as build_core)
    {
        double S[D][D], M[D][D];
        std::memcpy(S, sw.Lss, sizeof S);
        std::memcpy(M, sw.Gss, sizeof M);
        bool done = false;
        for (int it = 0; it < 48 && !done; ++it) {
            double u1[D][D], u2[D][D], M2[D][D], Mt[D][D];
            double mmax = 0.0;
            for (int i = 0; i < D; ++i)
                for (int j = 0; j < D; ++j) mmax = std::max(mmax, std::fabs(M[i][j]));
            mm<D>(M, S, u1);
            transpose<D>(M, Mt);
            mm<D>(u1, Mt, u2);
            mm<D>(M, M, M2);
            for (int i = 0; i < D; ++i)
                for (int j = i; j < D; ++j) {
                    const double v = S[i][j] + 0.5 * (u2[i][j] + u2[j][i]);
                    S[i][j] = S[j][i] = v;
                }
            std::memcpy(M, M2, sizeof M);
            done = mmax < 1e-10;
        }
        if (!done) {
            sp.why = kSlowMixing;
            return;
        }
        std::memcpy(sw.Psinf, S, sizeof S);
    }
    sp.vb = quad_sym<D>(hv, sw.Psinf);
    // the smoothed variances backwards from the final filtered state (build_tables_variances): tvb_j = sum_{k < j} g_k L g_k' + g_j Pss g_j', g_k = h' G^k
    {
        int n1 = -1;
        double g[D], Lsym[D][D], Psym[D][D];
        for (int i = 0; i < D; ++i) {
            g[i] = hv[i];
            for (int j = 0; j < D; ++j) {
                Lsym[i][j] = (i <= j) ? sw.Lss[i][j] : sw.Lss[j][i];
                Psym[i][j] = Pf[i][j];
            }
        }
        auto quad = [&](const double (&M)[D][D], const double (&x)[D]) {
            double s = 0.0;
            for (int i = 0; i < D; ++i) {
                double v = 0.0;
                for (int j = 0; j < D; ++j) v += M[i][j] * x[j];
                s += x[i] * v;
            }
            return s;
        };
        double acc = 0.0, prev = 0.0, prev2 = 0.0;
        for (int jt = 0; jt < kTailMax; ++jt) {
            const double v = acc + quad(Psym, g);
            tvb[jt] = v;
            if (jt >= 1 && (std::fabs(v - prev) <= kTol * std::fabs(v) || (jt >= 2 && v == prev2)) && std::fabs(v - sp.vb) <= 1e-9 * std::fabs(sp.vb)) {
                n1 = jt + 1;
                break;
            }
            prev2 = prev;
            prev = v;
            acc += quad(Lsym, g);
            double ng[D];
            for (int j = 0; j < D; ++j) ng[j] = 0.0;
            for (int i = 0; i < D; ++i) {
                const double f = g[i];
                for (int j = 0; j < D; ++j) ng[j] += f * sw.Gss[i][j];
            }
            for (int j = 0; j < D; ++j) g[j] = ng[j];
        }
        if (n1 < 0) {
            sp.why = kTailLong;
            return;
        }
        sp.n1 = n1;
        if ((long long)fp.nhs + n1 + 16 > T) {
            sp.why = kTooShort;
            return;
        }
    }
    // dense powers of G, its halo; the rows WJ, WG
    auto put = [](const double (&M)[D][D], double* out) {
        for (int i = 0; i < D; ++i)
            for (int k = 0; k < D; ++k) out[i * D + k] = M[i][k];
    };
    double X[D][D], Y[D][D], P16[D][D];
    std::memcpy(X, sw.Gss, sizeof X);
    for (int q = 0; q < 3; ++q) {
        mm<D>(X, X, Y);
        std::memcpy(X, Y, sizeof X);
    }
    for (int k = 0; k < 8; ++k) {
        if (k < 6) put(X, sp.GP[k]);
        else put(X, sp.GPT[k - 6]);
        if (k == 1) std::memcpy(P16, X, sizeof X);
        mm<D>(X, X, Y);
        std::memcpy(X, Y, sizeof X);
    }
    {
        double M[D][D];
        std::memcpy(M, P16, sizeof M);
        int n = 16;
        const double tiny = std::ldexp(1.0, -60);
        for (;;) {
            double mx = 0.0;
            for (int i = 0; i < D; ++i)
                for (int k = 0; k < D; ++k) mx = std::max(mx, std::fabs(M[i][k]));
            if (!std::isfinite(mx) || n > kHaloMax) {
                sp.why = kSlowMixing;
                return;
            }
            if (mx <= tiny) break;
            mm<D>(M, P16, Y);
            std::memcpy(M, Y, sizeof M);
            n += 16;
        }
        sp.halo = std::max(n, fp.halo);
    }
    {
        double x[D], nx[D];
        for (int i = 0; i < D; ++i) x[i] = hv[i];
        for (int j = 0; j < kSub; ++j) {      // row times Phi
            for (int i = 0; i < D; ++i) sp.WJ[j][i] = x[i];
            for (int k = 0; k < D; ++k) {
                double v = 0.0;
                for (int i = 0; i < D; ++i) v = pfma(x[i], fp.Phi[i * D + k], v);
                nx[k] = v;
            }
            for (int i = 0; i < D; ++i) x[i] = nx[i];
        }
        for (int i = 0; i < D; ++i) x[i] = hv[i];
        for (int j = kSub - 1; j >= 0; --j) {
            for (int i = 0; i < D; ++i) sp.WG[j][i] = x[i];
            for (int k = 0; k < D; ++k) {
                double v = 0.0;
                for (int i = 0; i < D; ++i) v = pfma(x[i], sw.Gss[i][k], v);
                nx[k] = v;
            }
            for (int i = 0; i < D; ++i) x[i] = nx[i];
        }
    }
}

// The head forwards (before the launch): y [nhs] -> the innovations (kept for the way back), the predicted mean of step nhs, sum r^2 / S_t
template <int D>
inline void smooth_head_forward(const ModelHost& m, const SmoothPlan& sp, const double* y, double* mu_end, double* quad, const double* hh_t = nullptr) {      // hh_t: the head's emission offsets when the model has one per step
    const FilterPlan& fp = sp.fp;
    const FilterWork<D>& fw = filter_work<D>();
    SmoothWork<D>& sw = smooth_work<D>();
    double mu[D], nm[D];
    for (int i = 0; i < D; ++i) {
        double v = m.a[i];
        for (int k = 0; k < D; ++k) v += fw.A[i][k] * m.x0m[k];
        mu[i] = v;
    }
    double q = 0.0;
    for (int t = 0; t < fp.nhs; ++t) {
        const int ti = t < fp.n0 ? t : fp.n0;
        double r = y[t] - (hh_t ? hh_t[t] : fp.hh);
        for (int k = 0; k < D; ++k) r -= fp.h[k] * mu[k];
        sw.r[t] = r;
        q += r * r * fw.iS[ti];
        for (int i = 0; i < D; ++i) {
            double v = m.a[i] + fw.kA[ti][i] * r;
            for (int k = 0; k < D; ++k) v += fw.A[i][k] * mu[k];
            nm[i] = v;
        }
        for (int i = 0; i < D; ++i) mu[i] = nm[i];
    }
    for (int i = 0; i < D; ++i) mu_end[i] = mu[i];
    *quad = q;
}
// The head's data-free tables (beside the kernel): G_t of every head step with gains of its own, the smoothed variances h' Ps_t h, t < nhs.
// false: a predicted covariance is not positive definite.
template <int D>
inline bool smooth_head_tables(const ModelHost& m, const SmoothPlan& sp) {
    using namespace detail;
    const FilterPlan& fp = sp.fp;
    const FilterWork<D>& fw = filter_work<D>();
    SmoothWork<D>& sw = smooth_work<D>();
    double A[D][D], At[D][D], Q[D][D], hv[D];
    for (int i = 0; i < D; ++i) {
        hv[i] = fp.h[i];
        for (int k = 0; k < D; ++k) {
            A[i][k] = fw.A[i][k];
            const int r = i < k ? i : k, c = i < k ? k : i;
            Q[i][k] = m.Q[r + c * D];
        }
    }
    transpose<D>(A, At);
    // G_t, t = 1 .. nhs - 1, takes step t to step t - 1: invert_dynamics(filtered t - 1, predicted t); settled once t - 1 >= n0
    const int tset = std::min(fp.nhs - 1, fp.n0 + 1);      // G_t = Gss for t >= tset
    double Pc[D][D];
    std::memcpy(Pc, sw.Psinf, sizeof Pc);
    for (int t = fp.nhs - 1; t >= tset; --t) {
        std::memcpy(sw.G[t], sw.Gss, sizeof sw.Gss);
        std::memcpy(sw.Ut[t], sw.Uss, sizeof sw.Uss);
        sw.vb[t] = sp.vb;
    }
    // (Ps_t = Psinf for t >= tset - 1: the steps behind are settled and the series' end is more than n1 steps away)
    for (int t = tset - 1; t >= 1; --t) {
        double Pf[D][D], Pp[D][D], t1[D][D], L[D][D], pn[D][D];
        for (int i = 0; i < D; ++i)
            for (int k = 0; k < D; ++k) Pf[i][k] = 0.5 * (fw.Pf[t - 1][i][k] + fw.Pf[t - 1][k][i]);
        mm<D>(A, Pf, t1);
        mm<D>(t1, At, Pp);
        for (int i = 0; i < D; ++i)
            for (int k = 0; k < D; ++k) Pp[i][k] += Q[i][k];
        if (!invert_dynamics<D>(A, Pf, Pp, sw.G[t], L)) return false;
        sw.u_ok = chol_upper<D>(L, 1e-9, sw.Ut[t]) && sw.u_ok;
        sw.vb[t] = quad_sym<D>(hv, Pc);
        smooth_cov_step<D>(sw.G[t], L, Pc, pn);
        std::memcpy(Pc, pn, sizeof pn);
    }
    sw.vb[0] = quad_sym<D>(hv, Pc);
    if (tset - 1 < 1 && fp.nhs >= 1) sw.vb[0] = sp.vb;
    return true;
}
// A draw from the posterior: what the kernel needs of the final filtering state's draw x_(T-1) = m_(T-1) + chol(P + 1e-12 I).U' eps_0
// (gaussian.jl:35-43) -- xi_T = U0' eps_0, v0 = G xi_T, s0 = h' xi_T -- and the settled noise factor U (row-major).  false: not positive definite.
template <int D>
inline bool smooth_rand_factors(const SmoothPlan& sp, const double* eps0, double* U_out, double* v0, double* s0) {
    const SmoothWork<D>& sw = smooth_work<D>();
    double P[D][D], U0[D][D], xiT[D];
    for (int i = 0; i < D; ++i)
        for (int k = 0; k < D; ++k) P[i][k] = sp.fp.Pss[i * D + k];
    if (!sw.u_ok || !chol_upper<D>(P, 1e-12, U0)) return false;
    for (int i = 0; i < D; ++i) {
        double v = 0.0;
        for (int k = 0; k <= i; ++k) v += U0[k][i] * eps0[k];
        xiT[i] = v;
    }
    double s = 0.0;
    for (int i = 0; i < D; ++i) {
        double v = 0.0;
        for (int k = 0; k < D; ++k) v += sw.Gss[i][k] * xiT[k];
        v0[i] = v;
        s += sp.fp.h[i] * xiT[i];
        for (int k = 0; k < D; ++k) U_out[i * D + k] = sw.Uss[i][k];
    }
    *s0 = s;
    return true;
}
// The head of a draw, backwards (behind the kernel): delta = x - m of step nhs - 1 (the kernel's xi at step nhs), the head's draws eps_e [nhs],
// eps_t [nhs][D], its emission noise variances rn (one, or nhs with rn_per_step) -> y_draw [nhs]
template <int D>
inline void smooth_head_backward_rand(const ModelHost& m, const SmoothPlan& sp, const double* y, const double* delta_in, const double* eps_e, const double* eps_t,
                                      const double* rn, bool rn_per_step, double* out) {
    const FilterPlan& fp = sp.fp;
    const FilterWork<D>& fw = filter_work<D>();
    const SmoothWork<D>& sw = smooth_work<D>();
    double dl[D], nl[D];
    for (int i = 0; i < D; ++i) dl[i] = delta_in[i];
    const double R = m.R[0];
    for (int t = fp.nhs - 1; t >= 0; --t) {
        const int ti = t < fp.n0 ? t : fp.n0;
        double o = y[t] - R * fw.iS[ti] * sw.r[t];
        for (int k = 0; k < D; ++k) o += fp.h[k] * dl[k];
        out[t] = o + std::sqrt(rn[rn_per_step ? t : 0]) * eps_e[t];
        if (t == 0) break;
        double w[D];
        for (int k = 0; k < D; ++k) w[k] = fw.K[ti][k] * sw.r[t] + dl[k];
        for (int i = 0; i < D; ++i) {
            double v = 0.0;
            for (int k = 0; k < D; ++k) v += sw.G[t][i][k] * w[k];
            for (int k = 0; k <= i; ++k) v += sw.Ut[t][k][i] * eps_t[(size_t)t * D + k];
            nl[i] = v;
        }
        for (int i = 0; i < D; ++i) dl[i] = nl[i];
    }
}

// The head backwards (behind the kernel): lam = xs - m of step nhs - 1 (the kernel's xi at step nhs) -> mean [nhs], vb [nhs] = h' Ps_t h
template <int D>
inline void smooth_head_backward(const ModelHost& m, const SmoothPlan& sp, const double* y, const double* lam_in, double* mean, double* vb) {
    const FilterPlan& fp = sp.fp;
    const FilterWork<D>& fw = filter_work<D>();
    const SmoothWork<D>& sw = smooth_work<D>();
    double lam[D], nl[D];
    for (int i = 0; i < D; ++i) lam[i] = lam_in[i];
    const double R = m.R[0];
    for (int t = fp.nhs - 1; t >= 0; --t) {
        const int ti = t < fp.n0 ? t : fp.n0;
        double o = y[t] - R * fw.iS[ti] * sw.r[t];
        for (int k = 0; k < D; ++k) o += fp.h[k] * lam[k];
        mean[t] = o;
        vb[t] = sw.vb[t];
        if (t == 0) break;
        double w[D];
        for (int k = 0; k < D; ++k) w[k] = fw.K[ti][k] * sw.r[t] + lam[k];
        for (int i = 0; i < D; ++i) {
            double v = 0.0;
            for (int k = 0; k < D; ++k) v += sw.G[t][i][k] * w[k];
            nl[i] = v;
        }
        for (int i = 0; i < D; ++i) lam[i] = nl[i];
    }
}

#define TGP_PLAN_DISPATCH(d, expr)                                 \
    switch (d) {                                                   \
        case 1: { constexpr int D = 1; return expr; }              \
        case 2: { constexpr int D = 2; return expr; }              \
        case 3: { constexpr int D = 3; return expr; }              \
        case 4: { constexpr int D = 4; return expr; }              \
        case 5: { constexpr int D = 5; return expr; }              \
        case 6: { constexpr int D = 6; return expr; }              \
        case 7: { constexpr int D = 7; return expr; }              \
        case 8: { constexpr int D = 8; return expr; }              \
    }
inline Info build_core_any(const ModelHost& m, long long T, Modal& md, HeadTables& tab) {
    TGP_PLAN_DISPATCH(m.d, build_core<D>(m, T, md, tab))
    Info bad;
    bad.why = kEigFail;
    return bad;
}
inline unsigned long long work_stamp_any(int d) {      // the stamp of this thread's workspace for state dimension d (0: never filled)
    TGP_PLAN_DISPATCH(d, work<D>().stamp)
    return 0;
}
inline int build_tables_any(int d, long long T, Modal& md, HeadTables& tab, Info& info) {
    TGP_PLAN_DISPATCH(d, build_tables<D>(T, md, tab, info))
    return kEigFail;
}
inline int build_tables_tail_any(int d, long long T, Modal& md, HeadTables& tab, Info& info) {
    TGP_PLAN_DISPATCH(d, build_tables_tail<D>(T, md, tab, info))
    return kEigFail;
}
inline int build_tables_headvar_any(int d, Modal& md, HeadTables& tab) {
    TGP_PLAN_DISPATCH(d, build_tables_headvar<D>(md, tab))
    return kEigFail;
}
inline int build_tables_stage_any(int d, int stage, long long T, Modal& md, HeadTables& tab, Info& info) {
    TGP_PLAN_DISPATCH(d, build_tables_stage<D>(stage, T, md, tab, info))
    return kEigFail;
}
inline Info build_any(const ModelHost& m, long long T, Modal& md, HeadTables& tab) {
    TGP_PLAN_DISPATCH(m.d, build<D>(m, T, md, tab))
    Info bad;
    bad.why = kEigFail;
    return bad;
}
inline void modal_head_forward_any(const ModelHost& m, const Modal& md, const HeadTables& tab, const double* y, double* r_out, double* z0, double* quad) {
    TGP_PLAN_DISPATCH(m.d, modal_head_forward<D>(m, md, tab, y, r_out, z0, quad))
}
inline void modal_head_backward_any(const Modal& md, const HeadTables& tab, const double* y, const double* r, const double* zeta, double* mean, double* vb) {
    TGP_PLAN_DISPATCH(md.d, modal_head_backward<D>(md, tab, y, r, zeta, mean, vb))
}
inline void build_smooth_any(const ModelHost& m, long long T, SmoothPlan& sp, double* tvb, bool post = true) {
    switch (m.d) {
        case 1: build_smooth<1>(m, T, sp, tvb, post); return;
        case 2: build_smooth<2>(m, T, sp, tvb, post); return;
        case 3: build_smooth<3>(m, T, sp, tvb, post); return;
        case 4: build_smooth<4>(m, T, sp, tvb, post); return;
        case 5: build_smooth<5>(m, T, sp, tvb, post); return;
        case 6: build_smooth<6>(m, T, sp, tvb, post); return;
        case 7: build_smooth<7>(m, T, sp, tvb, post); return;
        case 8: build_smooth<8>(m, T, sp, tvb, post); return;
    }
    sp.why = kEigFail;
}
inline void smooth_head_forward_any(const ModelHost& m, const SmoothPlan& sp, const double* y, double* mu_end, double* quad, const double* hh_t = nullptr) {
    TGP_PLAN_DISPATCH(m.d, smooth_head_forward<D>(m, sp, y, mu_end, quad, hh_t))
}
inline bool smooth_head_tables_any(const ModelHost& m, const SmoothPlan& sp) {
    TGP_PLAN_DISPATCH(m.d, smooth_head_tables<D>(m, sp))
    return false;
}
inline void smooth_head_backward_any(const ModelHost& m, const SmoothPlan& sp, const double* y, const double* lam, double* mean, double* vb) {
    TGP_PLAN_DISPATCH(m.d, smooth_head_backward<D>(m, sp, y, lam, mean, vb))
}
inline bool smooth_rand_factors_any(const SmoothPlan& sp, const double* eps0, double* U_out, double* v0, double* s0) {
    TGP_PLAN_DISPATCH(sp.fp.d, smooth_rand_factors<D>(sp, eps0, U_out, v0, s0))
    return false;
}
inline void smooth_head_backward_rand_any(const ModelHost& m, const SmoothPlan& sp, const double* y, const double* delta, const double* eps_e, const double* eps_t,
                                          const double* rn, bool rn_per_step, double* out) {
    TGP_PLAN_DISPATCH(m.d, smooth_head_backward_rand<D>(m, sp, y, delta, eps_e, eps_t, rn, rn_per_step, out))
}
#undef TGP_PLAN_DISPATCH

}  // namespace tgp_plan
