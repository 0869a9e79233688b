// Host half of the adjoint gradient of logpdf for an LTI model served by the stationary-gain engine (tgp_steady.hip).
//
// The engine's logpdf is  l = -1/2 sum_t [log 2 pi + log S_i(t) + r_t^2 / S_i(t)],  r_t = y_t - hh - h' mu_t,
// mu_{t+1} = A mu_t + a + kA_i(t) r_t,  mu_0 = A x0m + a,  i(t) = min(t, n0), where (S_i, kA_i = A K_i), i = 0..n0, come from the
// covariance recursion (predict lgc.jl:46-52, update lgc.jl:247-257) started at x0P -- n0 steps to the stationary gain.  Reverse mode:
//   psi_t  = d l / d mu_t = A' psi_{t+1} - h rho_t,    rho_t = -r_t / S + kA' psi_{t+1}           (psi_T = 0)
//   dA += sum psi_{t+1} mu_t' + psi_0 x0m',  da = sum psi_{t+1} + psi_0,  dkA_i += psi_{t+1} r_t,  dh -= sum rho_t mu_t,  dhh -= sum rho_t,
//   dS_i += -1/2 (1 / S_i - r_t^2 / S_i^2),  dx0m = A' psi_0
// The T - nh steps behind the head (nh = 512 head tiles) arrive summed from the device (the record of tgp_steady::GradRec); the nh head
// steps are run here from y[0 .. nh) and the psi the device hands over at the head's end; then ONE reverse sweep through the n0 + 1 steps of
// the covariance recursion turns (dkA_i, dS_i) into contributions to dA, dQ, dh, dR, dx0P.  Cost on the host: O(nh d^2 + n0 d^3).
// The reference has no counterpart (its gradients are Mooncake's reverse mode over the sequential loop: bench/single_output_gps.jl:149-156).
#pragma once
#include <cstdint>
#include <vector>

namespace tgp_adjoint {

struct Out {
    double *gA, *ga, *gQ, *gH, *ghh, *gR, *gx0m, *gx0P;      // column-major matrices as in tgp_model_set; any may be nullptr
};

inline int record_size(int d) { return 3 * d * d + 8 * d + 8 + d * (d + 1) / 2; }

// rec: the engine's record (record_size(d) doubles); yh: the first nyh observations (nyh >= 512 * head tiles).  Returns 0, or 1 when the
// record says the engine did not apply / the arguments do not fit it.
// head_steps >= 0: the head's length in steps when it is not a whole number of 512-step tiles (the one-launch adjoint, tgp_modal::adjoint_lti)
inline int finish(int d, const double* rec, const double* yh, int64_t nyh, const Out& out, int64_t head_steps = -1) {
    const int DD = d * d, NS = DD + 3 * d + 2;
    const double *SA = rec, *Sa = rec + DD, *Sk = rec + DD + d, *Srm = rec + DD + 2 * d;
    const double Sr = rec[DD + 3 * d], SSQ = rec[DD + 3 * d + 1];
    const double *psi_nh = rec + NS, *meta = rec + NS + 2 * d, *md = meta + 4;
    const int64_t n0 = (int64_t)meta[0], th = (int64_t)meta[1], T = (int64_t)meta[2];
    if (meta[3] != 1.0 || n0 < 0 || (th < 1 && head_steps < 0)) return 1;
    const int64_t nh = head_steps >= 0 ? head_steps : th * 512;
    if (nyh < nh || nh > T) return 1;
    // model blocks, row-major copies
    std::vector<double> A(DD), Q(DD), a(d), h(d), x0m(d), P(DD);
    for (int i = 0; i < d; ++i) {
        for (int k = 0; k < d; ++k) {
            A[i * d + k] = md[i + k * d];
            Q[i * d + k] = md[DD + d + i + k * d];
        }
        a[i] = md[DD + i];
        h[i] = md[2 * DD + d + i];
    }
    const double hh = md[2 * DD + 2 * d], R = md[2 * DD + 2 * d + 1];
    const double* x0 = md + 2 * DD + 2 * d + 2;
    for (int i = 0; i < d; ++i) x0m[i] = x0[i];
    for (int c = 0; c < d; ++c)
        for (int r = 0; r <= c; ++r) P[r * d + c] = P[c * d + r] = x0[d + c * (c + 1) / 2 + r];
    auto matmul = [d](const double* X, const double* Y, double* Z, bool tX = false, bool tY = false) {      // Z = op(X) op(Y)
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j) {
                double s = 0.0;
                for (int k = 0; k < d; ++k) s += (tX ? X[k * d + i] : X[i * d + k]) * (tY ? Y[j * d + k] : Y[k * d + j]);
                Z[i * d + j] = s;
            }
    };
    // ---- forward covariance recursion, steps 0 .. n0
    const int64_t ns = n0 + 1;
    std::vector<double> Pprev(ns * DD), Pp(ns * DD), v(ns * d), S(ns), K(ns * d), kA(ns * d), t1(DD), t2(DD);
    for (int64_t t = 0; t < ns; ++t) {
        for (int e = 0; e < DD; ++e) Pprev[t * DD + e] = P[e];
        matmul(A.data(), P.data(), t1.data());
        matmul(t1.data(), A.data(), t2.data(), false, true);
        double* pp = &Pp[t * DD];
        for (int e = 0; e < DD; ++e) pp[e] = t2[e] + Q[e];
        double s = R;
        for (int i = 0; i < d; ++i) {
            double x = 0.0;
            for (int k = 0; k < d; ++k) x += pp[i * d + k] * h[k];
            v[t * d + i] = x;
            s += h[i] * x;
        }
        S[t] = s;
        for (int i = 0; i < d; ++i) K[t * d + i] = v[t * d + i] / s;
        for (int i = 0; i < d; ++i) {
            double x = 0.0;
            for (int k = 0; k < d; ++k) x += A[i * d + k] * K[t * d + k];
            kA[t * d + i] = x;
        }
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j) P[i * d + j] = pp[i * d + j] - v[t * d + i] * v[t * d + j] / s;
    }
    // ---- head: forward means and innovations
    std::vector<double> mus(nh * d), r(nh), mu(d), nm(d);
    for (int i = 0; i < d; ++i) {
        double x = a[i];
        for (int k = 0; k < d; ++k) x += A[i * d + k] * x0m[k];
        mu[i] = x;
    }
    for (int64_t t = 0; t < nh; ++t) {
        const int64_t ix = t < n0 ? t : n0;
        double rr = yh[t] - hh;
        for (int k = 0; k < d; ++k) {
            mus[t * d + k] = mu[k];
            rr -= h[k] * mu[k];
        }
        r[t] = rr;
        for (int i = 0; i < d; ++i) {
            double x = a[i] + kA[ix * d + i] * rr;
            for (int k = 0; k < d; ++k) x += A[i * d + k] * mu[k];
            nm[i] = x;
        }
        mu = nm;
    }
    // ---- accumulators, seeded with the device's sums over the steps behind the head (all of them at index n0)
    std::vector<double> bA(DD), ba(d), bQ(DD, 0.0), bh(d), bkA(ns * d, 0.0), bS(ns, 0.0), psi(d), np(d);
    double bhh, bR = 0.0;
    const double Sn = S[n0];
    for (int e = 0; e < DD; ++e) bA[e] = SA[e];
    for (int i = 0; i < d; ++i) {
        ba[i] = Sa[i];
        bkA[n0 * d + i] = Sk[i];
        double x = -Srm[i] / Sn;                                   // sum rho_t mu_t = -Srm / S + SA' kA
        for (int k = 0; k < d; ++k) x += SA[k * d + i] * kA[n0 * d + k];
        bh[i] = -x;
    }
    {
        double x = -Sr / Sn;                                       // sum rho_t = -Sr / S + kA . Sa
        for (int k = 0; k < d; ++k) x += kA[n0 * d + k] * Sa[k];
        bhh = -x;
    }
    bS[n0] = -0.5 * ((double)(T - nh) / Sn - SSQ / (Sn * Sn));
    // ---- head, backwards
    for (int i = 0; i < d; ++i) psi[i] = psi_nh[i];
    for (int64_t t = nh - 1; t >= 0; --t) {
        const int64_t ix = t < n0 ? t : n0;
        const double rr = r[t], St = S[ix];
        double rho = -rr / St;
        for (int i = 0; i < d; ++i) {
            bkA[ix * d + i] += psi[i] * rr;
            ba[i] += psi[i];
            rho += kA[ix * d + i] * psi[i];
            for (int k = 0; k < d; ++k) bA[i * d + k] += psi[i] * mus[t * d + k];
        }
        for (int k = 0; k < d; ++k) bh[k] -= rho * mus[t * d + k];
        bhh -= rho;
        bS[ix] += -0.5 * (1.0 / St - rr * rr / (St * St));
        for (int i = 0; i < d; ++i) {
            double x = -h[i] * rho;
            for (int k = 0; k < d; ++k) x += A[k * d + i] * psi[k];
            np[i] = x;
        }
        psi = np;
    }
    for (int i = 0; i < d; ++i) {
        ba[i] += psi[i];
        for (int k = 0; k < d; ++k) bA[i * d + k] += psi[i] * x0m[k];
    }
    if (out.gx0m)
        for (int i = 0; i < d; ++i) {
            double x = 0.0;
            for (int k = 0; k < d; ++k) x += A[k * d + i] * psi[k];
            out.gx0m[i] = x;
        }
    // ---- reverse sweep through the covariance recursion
    std::vector<double> bPf(DD, 0.0), bPp(DD), bv(d), bK(d), t3(DD);
    for (int64_t t = ns - 1; t >= 0; --t) {
        const double *pp = &Pp[t * DD], *vt = &v[t * d], *Kt = &K[t * d], *Pq = &Pprev[t * DD];
        const double St = S[t];
        double bSt = bS[t];
        // Pf = Pp - v v' / S
        for (int e = 0; e < DD; ++e) bPp[e] = bPf[e];
        double q = 0.0;
        for (int i = 0; i < d; ++i) {
            double x = 0.0;
            for (int k = 0; k < d; ++k) {
                x += (bPf[i * d + k] + bPf[k * d + i]) * vt[k];
                q += vt[i] * bPf[i * d + k] * vt[k];
            }
            bv[i] = -x / St;
        }
        bSt += q / (St * St);
        // kA = A K, K = v / S
        double kv = 0.0;
        for (int i = 0; i < d; ++i) {
            double x = 0.0;
            for (int k = 0; k < d; ++k) {
                x += A[k * d + i] * bkA[t * d + k];
                bA[i * d + k] += bkA[t * d + i] * Kt[k];
            }
            bK[i] = x;
        }
        for (int i = 0; i < d; ++i) {
            bv[i] += bK[i] / St;
            kv += bK[i] * vt[i];
        }
        bSt -= kv / (St * St);
        // S = h' v + R
        for (int i = 0; i < d; ++i) {
            bh[i] += bSt * vt[i];
            bv[i] += bSt * h[i];
        }
        bR += bSt;
        // v = Pp h
        for (int i = 0; i < d; ++i)
            for (int k = 0; k < d; ++k) {
                bPp[i * d + k] += bv[i] * h[k];
                bh[k] += pp[i * d + k] * bv[i];
            }
        // Pp = A Symmetric(P) A' + Q
        for (int e = 0; e < DD; ++e) bQ[e] += bPp[e];
        for (int i = 0; i < d; ++i)
            for (int k = 0; k < d; ++k) t3[i * d + k] = bPp[i * d + k] + bPp[k * d + i];
        matmul(t3.data(), A.data(), t1.data());
        matmul(t1.data(), Pq, t2.data());
        for (int e = 0; e < DD; ++e) bA[e] += t2[e];
        matmul(A.data(), bPp.data(), t1.data(), true, false);
        matmul(t1.data(), A.data(), bPf.data());
    }
    // ---- outputs (column-major; symmetric blocks symmetrised)
    for (int i = 0; i < d; ++i)
        for (int k = 0; k < d; ++k) {
            if (out.gA) out.gA[i + k * d] = bA[i * d + k];
            if (out.gQ) out.gQ[i + k * d] = 0.5 * (bQ[i * d + k] + bQ[k * d + i]);
            if (out.gx0P) out.gx0P[i + k * d] = 0.5 * (bPf[i * d + k] + bPf[k * d + i]);
        }
    for (int i = 0; i < d; ++i) {
        if (out.ga) out.ga[i] = ba[i];
        if (out.gH) out.gH[i] = bh[i];
    }
    if (out.ghh) *out.ghh = bhh;
    if (out.gR) *out.gR = bR;
    return 0;
}

}  // namespace tgp_adjoint
