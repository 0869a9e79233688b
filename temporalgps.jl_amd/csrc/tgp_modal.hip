// Stationary-gain engine, ONE-LAUNCH path (round 4): logpdf and posterior marginals of an LTI model with one noise variance, scalar
// observations and no missing data in a single kernel over y -- see tgp_modal.hpp.  gfx950 only (wave64, __shfl scans, LDS rows for
// whole-line output stores, uniform coefficients through the kernel-argument segment).
#include "tgp_modal.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace tgp_modal {

namespace {

using tgp_plan::HeadTables;
using tgp_plan::Modal;

constexpr int kTile = 512, kSub = 8;
constexpr int kTileRow = kTile + 32;      // padded LDS row of a tile's outputs (see flush_row)
constexpr double kLog2Pi = 1.8378770664093454835606594728112;
typedef double v2d __attribute__((ext_vector_type(2)));

// ---- kernel arguments: every coefficient is wave-uniform and reaches the lanes through scalar loads ---------------------------------
template <int D>
struct KArgs {
    double fd[D], fo[D], fb[D], fa[D], fw[D];      // forward:  z' = fd z + fo z_partner + fb u + fa,  r = u - fw . z
    double gd[D], go[D], gc[D], gw[D];             // backward: zeta' = gd zeta + go zeta_partner + gc r,  mean = y - rS r + gw . zeta
    double fpr[6][D], fpi[6][D];                   // M^(8 2^k), k < 6 (forward block form: re, signed im)
    double gpr[6][D], gpi[6][D];
    double ftr[2][D], fti[2][D];                   // M^512, M^1024
    double gtr[2][D], gti[2][D];
    double WJ[kSub][D], WG[kSub][D];
    double hh, rS, vb;
    int n0, nhs, n1, halo, post, rnew_per_step;
    long long T, C, nwg;
    const double* y;
    const double* Rnew;
    double* mean;
    double* var;
    const HeadTables* tab;      // pinned host memory, read in place
    double* part;               // pinned host memory: [nwg] sum r^2 over the workgroups' core ranges, [nwg] the head's sum r^2 / S
};

template <int D>
__device__ __forceinline__ constexpr int partner(int i) {
    return ((i ^ 1) < D) ? (i ^ 1) : i;
}
// x <- P x with P the block form given by (pr, pi): (P x)_i = pr_i x_i + pi_i x_partner(i)
template <int D>
__device__ __forceinline__ void bmul(const double* __restrict__ pr, const double* __restrict__ pi, const double (&x)[D], double (&out)[D]) {
#pragma unroll
    for (int i = 0; i < D; ++i) out[i] = fma(pr[i], x[i], pi[i] * x[partner<D>(i)]);
}

__device__ __forceinline__ void lds_sync() {      // one wave talking to itself through LDS (DS operations of a wave execute in order)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ void load8(const double* __restrict__ p, long long t0, long long T, double (&v)[kSub]) {
    if (t0 + kSub <= T && t0 >= 0) {
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
        for (int j = 0; j < kSub; ++j) v[j] = (t0 + j < T && t0 + j >= 0) ? p[t0 + j] : 0.0;
    }
}

// A wave's tile of 512 consecutive output values, eight per lane, leaves through the wave's own LDS row: lane l's pairs go in at
// 16-byte slots 4 l + l / 4 + j / 2 (the pad keeps the 128-bit writes of sixteen lanes on distinct banks) and come out transposed, so
// that every store instruction writes 1 KB of consecutive bytes (as k_apply of tgp_steady.hip).  [lo, hi): the steps this workgroup owns.
__device__ __forceinline__ int tile_slot(int lane) { return lane * 4 + (lane >> 2); }
__device__ __forceinline__ void flush_row(double* __restrict__ p, long long tile_t0, long long lo, long long hi, const v2d* r2, int lane) {
    const bool aligned = (reinterpret_cast<uintptr_t>(p) & 15) == 0;
    if (tile_t0 >= lo && tile_t0 + kTile <= hi && aligned) {      // (wave-uniform)
        v2d* q = reinterpret_cast<v2d*>(p + tile_t0);
#pragma unroll
        for (int k = 0; k < kSub / 2; ++k) {
            const int e = k * 64 + lane, ls = e >> 2;
            q[e] = r2[ls * 4 + (ls >> 2) + (e & 3)];
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < kSub / 2; ++k) {
        const int e = k * 64 + lane, ls = e >> 2;
        const v2d w = r2[ls * 4 + (ls >> 2) + (e & 3)];
        const long long t = tile_t0 + 2 * e;
        if (t >= lo && t + 1 < hi && aligned) {
            *reinterpret_cast<v2d*>(p + t) = w;
        } else {
            if (t >= lo && t < hi) p[t] = w.x;
            if (t + 1 >= lo && t + 1 < hi) p[t + 1] = w.y;
        }
    }
}

// ---- the head: steps [0, nhs) with gains of their own, sequentially, by ONE wave (every lane the same arithmetic; the lanes share the
// loads and stores).  In the modal coordinates of the stationary closed loop the forward recursion costs O(d) per step:
//     z' = M z + fa + fb u + db_t r,   r = u - fw . z,   db_t = V^-1 (A K_t - A K)   (zero from step n0 on)
template <int D, int CH>
__device__ void head_forward(const KArgs<D>& ka, double* __restrict__ sY /*[kHeadMax]: y*/, double* __restrict__ sR /*[kHeadMax]: r*/,
                             double* __restrict__ sTab /*[CH (D + 1)]*/, int lane, double (&z0)[D], double& quad) {
    const HeadTables* __restrict__ tb = ka.tab;
    const int nhs = ka.nhs, n0 = ka.n0;
    for (int t = lane; t < nhs; t += 64) sY[t] = ka.y[t];
    double z[D];
#pragma unroll
    for (int i = 0; i < D; ++i) z[i] = tb->mu0[i];      // (already in modal coordinates: V^-1 (A x0.m + a))
    double acc = 0.0;
    for (int c0 = 0; c0 < nhs; c0 += CH) {
        lds_sync();
        for (int idx = lane; idx < CH * (D + 1); idx += 64) {
            const int s = idx / (D + 1), q = idx % (D + 1);
            const int t = c0 + s < n0 ? c0 + s : n0;
            sTab[idx] = (q < D) ? tb->kA[t * D + q] : tb->iS[t];
        }
        lds_sync();
        const int cend = (nhs - c0 < CH) ? nhs - c0 : CH;
        for (int s = 0; s < cend; ++s) {
            const double u = sY[c0 + s] - ka.hh;
            double r = u;
#pragma unroll
            for (int i = 0; i < D; ++i) r = fma(-ka.fw[i], z[i], r);
            acc = fma(r * r, sTab[s * (D + 1) + D], acc);
            double nz[D];
#pragma unroll
            for (int i = 0; i < D; ++i) {
                double v = fma(ka.fb[i], u, ka.fa[i]);
                v = fma(sTab[s * (D + 1) + i], r, v);
                v = fma(ka.fo[i], z[partner<D>(i)], v);
                nz[i] = fma(ka.fd[i], z[i], v);
            }
#pragma unroll
            for (int i = 0; i < D; ++i) z[i] = nz[i];
            if (lane == 0) sR[c0 + s] = r;
        }
    }
    lds_sync();
#pragma unroll
    for (int i = 0; i < D; ++i) z0[i] = z[i];
    quad = acc;
}

// Backward over the head from the lam in front of the first stationary step: lam <- G_t lam + c_t r_t (original coordinates: G_t is
// dense and changes per step), mean_t = y_t - (R / S_t) r_t + h . lam.  sR holds r_t on entry, the means on exit.
template <int D, int CH>
__device__ void head_backward(const KArgs<D>& ka, const double* __restrict__ sY, double* __restrict__ sR, double* __restrict__ sTab /*[CH (D D + D + 1)]*/, int lane,
                              const double (&zeta)[D]) {
    constexpr int DD = D * D, ROW = DD + D + 1;
    const HeadTables* __restrict__ tb = ka.tab;
    const int nhs = ka.nhs, n0 = ka.n0;
    double lam[D], h[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) v = fma(tb->Wm[i * D + k], zeta[k], v);
        lam[i] = v;
        h[i] = tb->h[i];
    }
    for (int hi = nhs; hi > 0; hi -= CH) {
        const int lo = hi - CH > 0 ? hi - CH : 0;
        lds_sync();
        for (int idx = lane; idx < (hi - lo) * ROW; idx += 64) {
            const int s = idx / ROW, q = idx % ROW;
            const int t = lo + s < n0 ? lo + s : n0;
            sTab[idx] = (q < DD) ? tb->G[(size_t)t * DD + q] : (q < DD + D ? tb->c[t * D + (q - DD)] : tb->rS[t]);
        }
        lds_sync();
        for (int t = hi - 1; t >= lo; --t) {
            const double* __restrict__ row = sTab + (t - lo) * ROW;
            const double r = sR[t];
            const double yv = sY[t];
            double m = fma(-row[DD + D], r, yv);
#pragma unroll
            for (int k = 0; k < D; ++k) m = fma(h[k], lam[k], m);
            double nl[D];
#pragma unroll
            for (int i = 0; i < D; ++i) {
                double v = row[DD + i] * r;
#pragma unroll
                for (int k = 0; k < D; ++k) v = fma(row[i * D + k], lam[k], v);
                nl[i] = v;
            }
#pragma unroll
            for (int i = 0; i < D; ++i) lam[i] = nl[i];
            if (lane == 0) sR[t] = m;
        }
    }
    lds_sync();
    const double rn0 = ka.Rnew[0];
    for (int t = lane; t < nhs; t += 64) {
        ka.mean[t] = sR[t];
        ka.var[t] = tb->vb[t < n0 ? t : n0] + (ka.rnew_per_step ? ka.Rnew[t] : rn0);
    }
}

// =================================================================================================================================
// The kernel.  Workgroup g owns the steps [nhs + g C, nhs + (g + 1) C) (its core); its NW waves are NW consecutive tiles of 512 steps that
// start `halo` steps earlier (g = 0: at nhs, with the head's exact end state) and end `halo` steps later.  Both mean recursions have
// forgotten a state after `halo` steps (tgp_steady_plan.hpp: |eigenvalue|^halo <= 2^-64), so a zero state at the start of the span and
// a zero lam at its end give every core step the same values, to rounding, as the recursion over the whole series: no pass over y
// before this one, no carries between workgroups.  Inside the workgroup the tiles are chained exactly (through LDS, over the at most
// three preceding / following tiles -- whatever lies further back has decayed as well).
// =================================================================================================================================
template <int D, int NW>
__global__ __launch_bounds__(NW * 64) void k_steady_one(const KArgs<D> ka) {
    constexpr int CHF = 32, CHB = (NW > 8 ? 8 : 16);
    constexpr int kStage = (CHB * (D * D + D + 1) > CHF * (D + 1)) ? CHB * (D * D + D + 1) : CHF * (D + 1);
    __shared__ double sF[NW][D], sB[NW][D], sHead[D], sAcc[NW];
    __shared__ double sY[tgp_plan::kHeadMax], sR[tgp_plan::kHeadMax], sTab[kStage];
    __shared__ __attribute__((aligned(16))) double sOut[NW][kTileRow];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // XCD-aware order: consecutive workgroups (which share their halos' lines of y) land on the same XCD, hence the same L2
    long long wg;
    {
        const long long per = (ka.nwg + 7) / 8;
        wg = (long long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
        if ((long long)(blockIdx.x >> 3) >= per || wg >= ka.nwg) return;
    }
    const long long T = ka.T;
    const long long c_lo = ka.nhs + wg * ka.C, c_hi_raw = c_lo + ka.C, c_hi = c_hi_raw < T ? c_hi_raw : T;
    const long long s0 = (wg == 0) ? (long long)ka.nhs : c_lo - ka.halo;
    const long long tile_t0 = s0 + (long long)wave * kTile, t0 = tile_t0 + (long long)lane * kSub;
    const bool head_wave = wg == 0 && wave == 0;
    const bool any_valid = tile_t0 < T;                              // (wave-uniform)
    const bool need_back = any_valid && tile_t0 + kTile > c_lo;      // a tile wholly inside the left halo only hands its end state on
    const bool has_out = ka.post && need_back && tile_t0 < c_hi;

    double head_quad = 0.0;
    if (head_wave) {
        double z0[D];
        head_forward<D, CHF>(ka, sY, sR, sTab, lane, z0, head_quad);
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < D; ++i) sHead[i] = z0[i];
        }
    }

    // ---- forward, zero start: innovations r0 of the lane's eight steps, the lane's end state, inclusive scan over the lanes
    double yv[kSub], r[kSub], st[D];
    const long long left = T - t0;
    const int nvalid = left >= kSub ? kSub : (left > 0 ? (int)left : 0);
    {
        double z[D];
#pragma unroll
        for (int i = 0; i < D; ++i) z[i] = 0.0;
        if (any_valid) {
            load8(ka.y, t0, T, yv);
#pragma unroll
            for (int j = 0; j < kSub; ++j) {
                const double u = yv[j] - ka.hh;
                double rr = u;
#pragma unroll
                for (int i = 0; i < D; ++i) rr = fma(-ka.fw[i], z[i], rr);
                r[j] = rr;
                double nz[D];
#pragma unroll
                for (int i = 0; i < D; ++i) nz[i] = fma(ka.fd[i], z[i], fma(ka.fo[i], z[partner<D>(i)], fma(ka.fb[i], u, ka.fa[i])));
#pragma unroll
                for (int i = 0; i < D; ++i) z[i] = nz[i];
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int off = 1 << k;
                double g[D], pg[D];
#pragma unroll
                for (int i = 0; i < D; ++i) g[i] = __shfl_up(z[i], off);
                bmul<D>(ka.fpr[k], ka.fpi[k], g, pg);
#pragma unroll
                for (int i = 0; i < D; ++i) z[i] = (lane >= off) ? z[i] + pg[i] : z[i];
            }
        }
#pragma unroll
        for (int i = 0; i < D; ++i) {
            const double up = __shfl_up(z[i], 1);
            st[i] = (lane == 0) ? 0.0 : up;
        }
        if (lane == 63) {
#pragma unroll
            for (int i = 0; i < D; ++i) sF[wave][i] = any_valid ? z[i] : 0.0;
        }
    }
    __syncthreads();
    double acc = 0.0;
    double zst[D];
#pragma unroll
    for (int i = 0; i < D; ++i) zst[i] = 0.0;
    if (any_valid) {
        // the tile's start state from the (at most three) tiles before it; workgroup 0: the head's end state sits at position -1
        double zin[D];
#pragma unroll
        for (int i = 0; i < D; ++i) zin[i] = 0.0;
#pragma unroll
        for (int k = 1; k <= 3; ++k) {
            const int src = wave - k;
            if (src < -1 || (src == -1 && wg != 0)) continue;      // (wave-uniform)
            double x[D];
#pragma unroll
            for (int i = 0; i < D; ++i) x[i] = (src >= 0) ? sF[src][i] : sHead[i];
            if (k == 1) {
#pragma unroll
                for (int i = 0; i < D; ++i) zin[i] += x[i];
            } else {
                double px[D];
                bmul<D>(ka.ftr[k - 2], ka.fti[k - 2], x, px);
#pragma unroll
                for (int i = 0; i < D; ++i) zin[i] += px[i];
            }
        }
        // the lane's start state: st + M^(8 lane) zin, by the bits of the lane number
        {
            double x[D];
#pragma unroll
            for (int i = 0; i < D; ++i) x[i] = zin[i];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                double px[D];
                bmul<D>(ka.fpr[k], ka.fpi[k], x, px);
#pragma unroll
                for (int i = 0; i < D; ++i) x[i] = ((lane >> k) & 1) ? px[i] : x[i];
            }
#pragma unroll
            for (int i = 0; i < D; ++i) st[i] += x[i];
        }
        const bool in_core = t0 >= c_lo && t0 < c_hi_raw;
#pragma unroll
        for (int j = 0; j < kSub; ++j) {
            double rr = r[j];
#pragma unroll
            for (int i = 0; i < D; ++i) rr = fma(-ka.WJ[j][i], st[i], rr);
            rr = (j < nvalid) ? rr : 0.0;
            r[j] = rr;
            acc = fma(rr, rr, acc);
        }
        acc = in_core ? acc : 0.0;
    }
    if (!ka.post) {
        // logpdf only: the sum of squares is all that is wanted
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) acc += __shfl_down(acc, off);
        if (lane == 0) sAcc[wave] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) t += sAcc[w];
            ka.part[wg] = t;
            if (wg == 0) ka.part[ka.nwg] = head_quad;
        }
        return;
    }
    // ---- backward, zero lam behind the tile: m0_j = y_j - rS r_j + gw . zeta (zeta from the lane's own later steps), reverse scan
    if (need_back) {
        double zeta[D];
#pragma unroll
        for (int i = 0; i < D; ++i) zeta[i] = 0.0;
#pragma unroll
        for (int j = kSub - 1; j >= 0; --j) {
            double m = fma(-ka.rS, r[j], yv[j]);
#pragma unroll
            for (int i = 0; i < D; ++i) m = fma(ka.gw[i], zeta[i], m);
            yv[j] = m;
            double nz[D];
#pragma unroll
            for (int i = 0; i < D; ++i) nz[i] = fma(ka.gd[i], zeta[i], fma(ka.go[i], zeta[partner<D>(i)], ka.gc[i] * r[j]));
#pragma unroll
            for (int i = 0; i < D; ++i) zeta[i] = nz[i];
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int off = 1 << k;
            double g[D], pg[D];
#pragma unroll
            for (int i = 0; i < D; ++i) g[i] = __shfl_down(zeta[i], off);
            bmul<D>(ka.gpr[k], ka.gpi[k], g, pg);
#pragma unroll
            for (int i = 0; i < D; ++i) zeta[i] = (lane + off < 64) ? zeta[i] + pg[i] : zeta[i];
        }
#pragma unroll
        for (int i = 0; i < D; ++i) {
            const double dn = __shfl_down(zeta[i], 1);
            zst[i] = (lane == 63) ? 0.0 : dn;
        }
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < D; ++i) sB[wave][i] = zeta[i];
        }
    } else if (lane == 0) {
#pragma unroll
        for (int i = 0; i < D; ++i) sB[wave][i] = 0.0;
    }
    __syncthreads();
    // the lam entering a tile from the right: from the (at most three) tiles behind it
    auto right_input = [&](int w, double (&zin)[D]) {
#pragma unroll
        for (int i = 0; i < D; ++i) zin[i] = 0.0;
#pragma unroll
        for (int k = 1; k <= 3; ++k) {
            const int src = w + k;
            if (src >= NW) continue;
            double x[D];
#pragma unroll
            for (int i = 0; i < D; ++i) x[i] = sB[src][i];
            if (k == 1) {
#pragma unroll
                for (int i = 0; i < D; ++i) zin[i] += x[i];
            } else {
                double px[D];
                bmul<D>(ka.gtr[k - 2], ka.gti[k - 2], x, px);
#pragma unroll
                for (int i = 0; i < D; ++i) zin[i] += px[i];
            }
        }
    };
    if (has_out) {
        double zin[D];
        right_input(wave, zin);
        {
            double x[D];
#pragma unroll
            for (int i = 0; i < D; ++i) x[i] = zin[i];
            const int back = 63 - lane;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                double px[D];
                bmul<D>(ka.gpr[k], ka.gpi[k], x, px);
#pragma unroll
                for (int i = 0; i < D; ++i) x[i] = ((back >> k) & 1) ? px[i] : x[i];
            }
#pragma unroll
            for (int i = 0; i < D; ++i) zst[i] += x[i];
        }
        v2d* row = reinterpret_cast<v2d*>(sOut[wave]);
        const int wb = tile_slot(lane);
#pragma unroll
        for (int j = 0; j < kSub; j += 2) {
            double m0 = yv[j], m1 = yv[j + 1];
#pragma unroll
            for (int i = 0; i < D; ++i) {
                m0 = fma(ka.WG[j][i], zst[i], m0);
                m1 = fma(ka.WG[j + 1][i], zst[i], m1);
            }
            v2d w;
            w.x = m0;
            w.y = m1;
            row[wb + (j >> 1)] = w;
        }
        lds_sync();
        flush_row(ka.mean, tile_t0, c_lo, c_hi, row, lane);
        // the variances do not depend on the data: a constant outside the last n1 steps (plus the new noise); written transposed as well
        {
            const double rn0 = ka.Rnew[0];
            const long long n1 = ka.n1;
            const double* __restrict__ tvb = ka.tab->tvb;
            const bool aligned = (reinterpret_cast<uintptr_t>(ka.var) & 15) == 0 && (!ka.rnew_per_step || (reinterpret_cast<uintptr_t>(ka.Rnew) & 15) == 0);
#pragma unroll
            for (int k = 0; k < kSub / 2; ++k) {
                const long long t = tile_t0 + 2 * (k * 64 + lane);
                if (t + 1 < c_lo || t >= c_hi) continue;
                double v0 = ka.vb, v1 = ka.vb;
                const long long b0 = T - 1 - t, b1 = b0 - 1;
                if (b0 < n1) {                           // (the last tiles only)
                    if (b0 >= 0) v0 = tvb[b0];
                    if (b1 >= 0) v1 = tvb[b1];
                } else if (b1 < n1 && b1 >= 0) {
                    v1 = tvb[b1];
                }
                if (t >= c_lo && t + 1 < c_hi && aligned) {
                    v2d rn;
                    if (ka.rnew_per_step) {
                        rn = *reinterpret_cast<const v2d*>(ka.Rnew + t);
                    } else {
                        rn.x = rn0;
                        rn.y = rn0;
                    }
                    v2d w;
                    w.x = v0 + rn.x;
                    w.y = v1 + rn.y;
                    *reinterpret_cast<v2d*>(ka.var + t) = w;
                } else {
                    if (t >= c_lo && t < c_hi) ka.var[t] = v0 + (ka.rnew_per_step ? ka.Rnew[t] : rn0);
                    if (t + 1 >= c_lo && t + 1 < c_hi) ka.var[t + 1] = v1 + (ka.rnew_per_step ? ka.Rnew[t + 1] : rn0);
                }
            }
        }
    }
    if (head_wave) {
        double zin[D];
        right_input(-1, zin);
        head_backward<D, CHB>(ka, sY, sR, sTab, lane, zin);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_down(acc, off);
    if (lane == 0) sAcc[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += sAcc[w];
        ka.part[wg] = t;
        if (wg == 0) ka.part[ka.nwg] = head_quad;
    }
}

template <int D>
void fill_args(KArgs<D>& ka, const Modal& md) {
    for (int i = 0; i < D; ++i) {
        ka.fd[i] = md.fd[i]; ka.fo[i] = md.fo[i]; ka.fb[i] = md.fb[i]; ka.fa[i] = md.fa[i]; ka.fw[i] = md.fw[i];
        ka.gd[i] = md.gd[i]; ka.go[i] = md.go[i]; ka.gc[i] = md.gc[i]; ka.gw[i] = md.gw[i];
        ka.fpr[0][i] = md.fp8r[i]; ka.fpi[0][i] = md.fp8i[i];
        ka.gpr[0][i] = md.gp8r[i]; ka.gpi[0][i] = md.gp8i[i];
        ka.ftr[0][i] = md.fp512r[i]; ka.fti[0][i] = md.fp512i[i];
        ka.gtr[0][i] = md.gp512r[i]; ka.gti[0][i] = md.gp512i[i];
        // squares (the sign convention of the imaginary parts is preserved: re^2 - im^2, 2 re im)
        for (int k = 1; k < 6; ++k) {
            ka.fpr[k][i] = ka.fpr[k - 1][i] * ka.fpr[k - 1][i] - ka.fpi[k - 1][i] * ka.fpi[k - 1][i];
            ka.fpi[k][i] = 2.0 * ka.fpr[k - 1][i] * ka.fpi[k - 1][i];
            ka.gpr[k][i] = ka.gpr[k - 1][i] * ka.gpr[k - 1][i] - ka.gpi[k - 1][i] * ka.gpi[k - 1][i];
            ka.gpi[k][i] = 2.0 * ka.gpr[k - 1][i] * ka.gpi[k - 1][i];
        }
        ka.ftr[1][i] = ka.ftr[0][i] * ka.ftr[0][i] - ka.fti[0][i] * ka.fti[0][i];
        ka.fti[1][i] = 2.0 * ka.ftr[0][i] * ka.fti[0][i];
        ka.gtr[1][i] = ka.gtr[0][i] * ka.gtr[0][i] - ka.gti[0][i] * ka.gti[0][i];
        ka.gti[1][i] = 2.0 * ka.gtr[0][i] * ka.gti[0][i];
        for (int j = 0; j < kSub; ++j) {
            ka.WJ[j][i] = md.WJ[j][i];
            ka.WG[j][i] = md.WG[j][i];
        }
    }
    ka.hh = md.hh; ka.rS = md.rS; ka.vb = md.vb;
    ka.n0 = md.n0; ka.nhs = md.nhs; ka.n1 = md.n1; ka.halo = md.halo;
}

}  // namespace

struct Engine {
    HeadTables* tab = nullptr;      // pinned host memory
    double* part = nullptr;         // pinned host memory
    size_t part_cap = 0;
    Modal md{};
    tgp_plan::Info info{};
    long long nwg = 0;
    int nw = 8;
    bool began = false;
};

Engine* create() { return new Engine(); }
void destroy(Engine* e) {
    if (!e) return;
    if (e->tab) (void)hipHostFree(e->tab);
    if (e->part) (void)hipHostFree(e->part);
    delete e;
}

const tgp_plan::Info& last_plan(const Engine* e) { return e->info; }
const tgp_plan::Modal& last_modal(const Engine* e) { return e->md; }

namespace {
template <int D>
int launch(Engine* e, hipStream_t st, const Call& c, const char** kname) {
    KArgs<D> ka;
    std::memset(&ka, 0, sizeof ka);
    fill_args<D>(ka, e->md);
    ka.post = c.mean != nullptr ? 1 : 0;
    ka.rnew_per_step = c.rnew_per_step;
    ka.T = c.T;
    const int nw = e->nw;
    ka.C = (long long)nw * kTile - 2LL * e->md.halo;
    ka.nwg = e->nwg;
    ka.y = c.y;
    ka.Rnew = c.Rnew;
    ka.mean = c.mean;
    ka.var = c.var;
    ka.tab = e->tab;
    ka.part = e->part;
    const long long per = (e->nwg + 7) / 8;
    const unsigned grid = (unsigned)(per * 8);
    if (nw == 8) {
        *kname = ka.post ? "k_steady_one<posterior>" : "k_steady_one<logpdf>";
        hipLaunchKernelGGL((k_steady_one<D, 8>), dim3(grid), dim3(8 * 64), 0, st, ka);
    } else {
        *kname = ka.post ? "k_steady_one16<posterior>" : "k_steady_one16<logpdf>";
        hipLaunchKernelGGL((k_steady_one<D, 16>), dim3(grid), dim3(16 * 64), 0, st, ka);
    }
    return (int)hipGetLastError();
}
}  // namespace

bool plan(Engine* e, const tgp_plan::ModelHost& m, long long T) {
    e->began = false;
    if (!e->tab && hipHostMalloc(reinterpret_cast<void**>(&e->tab), sizeof(HeadTables), hipHostMallocDefault) != hipSuccess) {
        e->tab = nullptr;
        e->info = tgp_plan::Info{};
        e->info.why = tgp_plan::kEigFail;
        return false;
    }
    e->info = tgp_plan::build_any(m, T, e->md, *e->tab);
    if (e->info.why != tgp_plan::kOk) return false;
    // workgroups of 8 tiles unless the halos would eat more than ~30 % of them; then 16
    const int halo = e->md.halo;
    e->nw = (2 * halo * 10 <= 3 * 8 * kTile) ? 8 : 16;
    const long long C = (long long)e->nw * kTile - 2LL * halo;
    e->nwg = (T - e->md.nhs + C - 1) / C;
    const size_t need = (size_t)e->nwg + 8;
    if (need > e->part_cap) {
        if (e->part) (void)hipHostFree(e->part);
        e->part = nullptr;
        e->part_cap = 0;
        if (hipHostMalloc(reinterpret_cast<void**>(&e->part), need * sizeof(double), hipHostMallocDefault) != hipSuccess) {
            e->info.why = tgp_plan::kEigFail;
            return false;
        }
        e->part_cap = need;
    }
    e->began = true;
    return true;
}

int enqueue(Engine* e, hipStream_t stream, const Call& c, const char** kname, std::string* err) {
    if (!e || !e->began || !c.y || c.T <= 0 || (c.mean && (!c.var || !c.Rnew))) {
        if (err) *err = "tgp_modal::enqueue: bad argument / no plan";
        return (int)hipErrorInvalidValue;
    }
    int r = 0;
    switch (e->md.d) {
        case 1: r = launch<1>(e, stream, c, kname); break;
        case 2: r = launch<2>(e, stream, c, kname); break;
        case 3: r = launch<3>(e, stream, c, kname); break;
        case 4: r = launch<4>(e, stream, c, kname); break;
        case 5: r = launch<5>(e, stream, c, kname); break;
        case 6: r = launch<6>(e, stream, c, kname); break;
        case 7: r = launch<7>(e, stream, c, kname); break;
        case 8: r = launch<8>(e, stream, c, kname); break;
        default: r = (int)hipErrorInvalidValue;
    }
    if (r != 0 && err) *err = std::string("tgp_modal: launch: ") + hipGetErrorString((hipError_t)r);
    return r;
}

// After the stream has passed the kernel: the log marginal likelihood from the workgroups' sums (fixed order).
double finish(const Engine* e, long long T) {
    const Modal& md = e->md;
    double s = 0.0;
    for (long long g = 0; g < e->nwg; ++g) s += e->part[g];
    const double quad = e->part[e->nwg] + md.iS * s;
    const double logdet = md.LS + (double)(T - md.n0) * md.logS;
    return -0.5 * ((double)T * kLog2Pi + logdet + quad);
}

}  // namespace tgp_modal
