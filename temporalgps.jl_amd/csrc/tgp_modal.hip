// Stationary-gain engine, ONE-LAUNCH path (round 4): logpdf and posterior marginals of an LTI model with one noise variance, scalar
// observations and no missing data in a single kernel over y -- see tgp_modal.hpp.  gfx950 only (wave64, __shfl scans, LDS rows for
// whole-line output stores, uniform coefficients through the kernel-argument segment).
#include "tgp_modal.hpp"
#include "tgp_alloc.hpp"
#include "tgp_lml.hpp"
#include "tgp_post.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace tgp_modal {

namespace {

using tgp_plan::HeadTables;
using tgp_plan::Modal;

constexpr int kWJ = 8;      // rows of the WJ / WG tables (a lane of sixteen steps applies them in two halves)
constexpr double kLog2Pi = 1.8378770664093454835606594728112;
typedef double v2d __attribute__((ext_vector_type(2)));

// ---- the packed tables of a call (doubles): [0] n1, then the fields at these offsets.  Per-step tables have n0 + 1 rows (row n0: the
// stationary step; the head's steps behind n0 read that row).
struct TabOff {
    int h, mu0, Wm, db, iS, rS, G, c, vb, tvb;
};

// ---- kernel arguments: every coefficient is wave-uniform and reaches the lanes through scalar loads ---------------------------------
template <int D>
struct KArgs {
    double fd[D], fo[D], fb[D], fa[D], fw[D];      // forward:  z' = fd z + fo z_partner + fb u + fa,  r = u - fw . z
    double gd[D], go[D], gc[D], gw[D];             // backward: zeta' = gd zeta + go zeta_partner + gc r,  mean = y - rS r + gw . zeta
    double fpr[6][D], fpi[6][D];                   // M^(SUB 2^k), k < 6 (forward block form: re, signed im; SUB = steps per lane)
    double gpr[6][D], gpi[6][D];
    double ftr[2][D], fti[2][D];                   // M^TILE, M^(2 TILE)
    double gtr[2][D], gti[2][D];
    double WJ[kWJ][D], WG[kWJ][D];
    double hh, rS, vb;
    int n0, nhs, halo, post, rnew_per_step;
    long long T, C, nwg, seq;
    // a time segment of a longer series (one rank of several; single device: wg0 = 0, seg = [0, T), T_eff = T): this launch owns the outputs
    // of [seg_lo, seg_hi) and runs the workgroups wg0 .. wg0 + nwg - 1 of the series; y, mean, var, Rnew (per step) are indexed by the
    // GLOBAL step (the pointers are moved back by seg_lo), the `halo` observations in front of / behind the segment come in yl / yr
    long long wg0, seg_lo, seg_hi, T_eff;
    const double *yl, *yr;
    const double* y;
    const double* Rnew;         // [0]: the shared new noise
    const double* RnewT;        // per-step new noise, indexed by the global step
    double* mean;
    double* var;
    const double* htab;         // pinned host memory: the packed head / tail tables of this call (TabOff), written by the host beside the kernel
    double* tab;                // device memory: workgroup 0 pulls them in here for its head wave
    const long long* flag;      // pinned host memory, one word per stage of the tables: 2 seq (+ 1: declined) once that stage is in `htab`
    TabOff to;
    double* part;               // pinned host memory: [nwg] sum r^2 over the workgroups' core ranges, [nwg] the head's sum r^2 / S
    // The head ON THE HOST (round 5): pinned host memory, flags hold 2 seq once raised.  Workgroup 0 hands the head's nhs observations
    // (and new noise variances) over first thing (hflag[0]), waits -- bounded -- for the head's end state z0 in modal coordinates (hflag[1]) where
    // it used to run the head forwards, hands back the backward state entering the head (zeta_out, hflag[2]) where it used to run it backwards,
    // and a workgroup from the middle of the dispatch writes the head's outputs once the host has them (head_out: nhs means, nhs variances; hflag[3]).
    int hosthead;
    double* head_in;
    const double* z0p;
    double* zeta_out;
    const double* head_out;
    long long* hflag;
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

// The tables half of the host plan (per-step gains of the head, tail variances) may still be in the making when the kernel starts: the host
// writes the packed tables into pinned memory and then the flag 2 seq (+ 1 if that half declined: the call is then re-run elsewhere and
// whatever this kernel writes is discarded).  Workgroup 0 pulls them into device memory, all its waves at once (one PCIe round trip with
// every load in flight; a DMA copy of the same bytes reaches the device ~20 us after the API call)
// The wait is BOUNDED (round-4 advice): a host thread that dies between the launch and the flag, or a tool that makes the launch synchronous,
// would otherwise leave the queue spinning for ever.  After kWaitTicks of the 100 MHz clock (two seconds: the host half takes microseconds)
// the kernel goes on with whatever the table memory holds -- finite garbage at worst, every index comes from the arguments -- and poisons
// its share of the sum of squares (`poison`, when given), so that the call's result is NaN and is discarded.
constexpr long long kWaitTicks = 200000000ll;
__device__ __noinline__ void wait_tables(const long long* flagc, long long seq, double* poison = nullptr) {
    long long* flag = const_cast<long long*>(flagc);
    const long long t0 = (long long)wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < 2 * seq) {
        __builtin_amdgcn_s_sleep(16);
        if ((long long)wall_clock64() - t0 > kWaitTicks) {
            if (poison != nullptr) *poison = __builtin_nan("");
            return;
        }
    }
}

template <int SUB, int D>
__device__ __forceinline__ void load_lane(const KArgs<D>& ka, long long t0, double (&v)[SUB]) {
    const double* __restrict__ p = ka.y;
    const long long hi = ka.seg_hi < ka.T_eff ? ka.seg_hi : ka.T_eff;
    if (t0 >= ka.seg_lo && t0 + SUB <= hi) {
        if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
            const double2* q = reinterpret_cast<const double2*>(p + t0);
#pragma unroll
            for (int j = 0; j < SUB / 2; ++j) {
                const double2 w = q[j];
                v[2 * j] = w.x;
                v[2 * j + 1] = w.y;
            }
        } else {
#pragma unroll
            for (int j = 0; j < SUB; ++j) v[j] = p[t0 + j];
        }
    } else {
        // (the ends of the series, and the few lanes of a segment's first / last workgroup that read its neighbours' observations)
#pragma unroll
        for (int j = 0; j < SUB; ++j) {
            const long long t = t0 + j;
            double x = 0.0;
            if (t < ka.T_eff) {
                if (t < ka.seg_lo) x = ka.yl[t - (ka.seg_lo - ka.halo)];
                else if (t >= ka.seg_hi) x = ka.yr[t - ka.seg_hi];
                else x = p[t];
            }
            v[j] = x;
        }
    }
}

// A wave's tile of 64 SUB consecutive output values, SUB per lane, leaves through the wave's own LDS row: lane l's SUB / 2 pairs go in at
// 16-byte slots PPL l + l PPL / 16 + j (the pad keeps the 128-bit writes of sixteen lanes on distinct banks) and come out transposed, so that
// every store instruction writes 1 KB of consecutive bytes (as k_apply of tgp_steady.hip).  [lo, hi): the steps this workgroup owns.
// The row holds HALF a tile (the values of 32 lanes: 2 KB + padding); a tile leaves in two rounds -- the LDS a full row would take
// (35 KB per workgroup) cost a workgroup per CU at d = 2, 4 (and would at d = 3 once its registers fit four).
template <int SUB>
struct Row {
    static constexpr int PPL = SUB / 2;                  // pairs per lane
    static constexpr int slots = PPL * 34;               // 16-byte slots of a row: the pairs of 32 lanes, padded
    __device__ static __forceinline__ int lane_slot(int l32) { return PPL * l32 + l32 * PPL / 16; }      // l32: the lane within its half
    __device__ static __forceinline__ int pair_slot(int e) {      // pair e of the half tile (time order)
        const int ls = e / PPL;
        return PPL * ls + ls * PPL / 16 + e % PPL;
    }
};
__device__ __forceinline__ void store_pair(v2d* q, const v2d w) { *q = w; }      // (non-temporal and write-through (sc0 sc1) stores: measured, no difference)
// half_t0: the first step of the half tile in the row
template <int SUB>
__device__ __forceinline__ void flush_row(double* __restrict__ p, long long half_t0, long long lo, long long hi, const v2d* r2, int lane) {
    constexpr int HALF = 32 * SUB;
    const bool aligned = (reinterpret_cast<uintptr_t>(p) & 15) == 0;
    if (half_t0 >= lo && half_t0 + HALF <= hi && aligned) {      // (wave-uniform)
        v2d* q = reinterpret_cast<v2d*>(p + half_t0);
#pragma unroll
        for (int k = 0; k < SUB / 4; ++k) {
            const int e = k * 64 + lane;
            store_pair(q + e, r2[Row<SUB>::pair_slot(e)]);
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < SUB / 4; ++k) {
        const int e = k * 64 + lane;
        const v2d w = r2[Row<SUB>::pair_slot(e)];
        const long long t = half_t0 + 2 * e;
        if (t >= lo && t + 1 < hi && aligned) {
            store_pair(reinterpret_cast<v2d*>(p + t), w);
        } else {
            if (t >= lo && t < hi) p[t] = w.x;
            if (t + 1 >= lo && t + 1 < hi) p[t + 1] = w.y;
        }
    }
}

// ---- the head: steps [0, nhs) with gains of their own, sequentially, by ONE wave.  Nothing in the dependent chain of a step waits for memory:
// the lanes fetch a block's per-step data ahead (lane l the data of step l of the block; the tables are read in place from pinned host memory, so a
// fetch costs a PCIe round trip -- it overlaps the block in work) and a step picks its values out of the lanes' registers with v_readlane.
__device__ __forceinline__ double readlane_d(double x, int l) {      // l wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), l), hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
    return __hiloint2double(hi, lo);
}

// Data-parallel-primitive moves of a double (two 32-bit v_mov_dpp: no LDS round trip as ds_bpermute has).  Lanes without a source, and the
// rows the mask leaves out, read zero.  gfx9 controls: row_shl:n 0x100 + n (lane l reads l + n of its row of 16), row_shr:n 0x110 + n
// (reads l - n), wave_shl:1 0x130 / wave_shr:1 0x138 (the same across the whole wave), row_bcast:15 0x142 (lane 15 of a row to the next row),
// row_bcast:31 0x143 (lane 31 to rows 2, 3), row_newbcast:n 0x150 + n (lane n of a row to its own row)
template <int CTRL, int ROWMASK = 0xF>
__device__ __forceinline__ double dpp_mov(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROWMASK, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROWMASK, 0xF, true);
    return __hiloint2double(hi, lo);
}
// the sum over the wave, in lane 63 (rows by row_shr adds, then row_bcast:15 / :31 across them)
__device__ __forceinline__ double wave_sum_to_lane63(double x) {
    x += dpp_mov<0x111>(x);
    x += dpp_mov<0x112>(x);
    x += dpp_mov<0x114>(x);
    x += dpp_mov<0x118>(x);
    x += dpp_mov<0x142, 0xA>(x);
    x += dpp_mov<0x143, 0xC>(x);
    return x;
}
// one level of the in-row scans: z += (pr, pi) (*) z shifted by OFF lanes inside the row of 16 (DIR 0: from below, 1: from above)
// (level K: OFF = 2^K lanes, the power M^(SUB 2^K) of the kernel arguments -- read member by member: a pointer into the by-value argument
//  struct would make the compiler keep a copy of it in scratch memory)
#define TGP_ROW_LEVEL(DIR, K, PR, PI, Z)                                                                          \
    do {                                                                                                          \
        double g_[D];                                                                                             \
        _Pragma("unroll") for (int i = 0; i < D; ++i) g_[i] = dpp_mov<((DIR) == 0 ? 0x110 : 0x100) + (1 << (K))>(Z[i]); \
        _Pragma("unroll") for (int i = 0; i < D; ++i) Z[i] = fma(ka.PR[K][i], g_[i], fma(ka.PI[K][i], g_[partner<D>(i)], Z[i])); \
    } while (0)

// Forward, in the modal coordinates of the stationary closed loop (O(d) per step; every lane the same arithmetic):
//     z' = M z + fa + fb u + db_t r,   r = u - fw . z,   db_t = V^-1 (A K_t - A K)   (zero from step n0 on)
// Leaves r_t in sR, the end state in z0, sum r^2 / S_t in quad.
template <int D>
__device__ __forceinline__ void head_forward(const KArgs<D>& ka, double* __restrict__ sR /*[kHeadMax]*/, int lane, double (&z0)[D], double& quad) {
    const double* __restrict__ tb = ka.tab;
    const int nhs = ka.nhs, n0 = ka.n0;
    double z[D];
#pragma unroll
    for (int i = 0; i < D; ++i) z[i] = tb[ka.to.mu0 + i];      // (already in modal coordinates: V^-1 (A x0.m + a))
    double acc = 0.0;
    const int nblk = (nhs + 63) >> 6;
    for (int b = 0; b < nblk; ++b) {
        // lane l: the data of step 64 b + l
        const int t = b * 64 + lane, ti = t < n0 ? t : n0;
        const double cy = t < nhs ? ka.y[t] : 0.0, cis = tb[ka.to.iS + ti];
        double cdb[D];
#pragma unroll
        for (int i = 0; i < D; ++i) cdb[i] = tb[ka.to.db + ti * D + i];
        const int cnt = nhs - b * 64 < 64 ? nhs - b * 64 : 64;
        double rk = 0.0;
        for (int s = 0; s < cnt; ++s) {
            const int su = __builtin_amdgcn_readfirstlane(s);
            const double u = readlane_d(cy, su) - ka.hh, is = readlane_d(cis, su);
            double r = u;
#pragma unroll
            for (int i = 0; i < D; ++i) r = fma(-ka.fw[i], z[i], r);
            acc = fma(r * r, is, acc);
            double nz[D];
#pragma unroll
            for (int i = 0; i < D; ++i) {
                double v = fma(ka.fb[i], u, ka.fa[i]);
                v = fma(readlane_d(cdb[i], su), r, v);
                v = fma(ka.fo[i], z[partner<D>(i)], v);
                nz[i] = fma(ka.fd[i], z[i], v);
            }
#pragma unroll
            for (int i = 0; i < D; ++i) z[i] = nz[i];
            rk = (lane == su) ? r : rk;
        }
        if (lane < cnt) sR[b * 64 + lane] = rk;
    }
    lds_sync();
#pragma unroll
    for (int i = 0; i < D; ++i) z0[i] = z[i];
    quad = acc;
}

// Backward over the head from the lam in front of the first stationary step: lam <- G_t lam + c_t r_t (original coordinates: G_t is dense and
// changes per step), mean_t = y_t - (R / S_t) r_t + h . lam.  Lane i < D owns row i of the product; lam goes round by v_readlane.  The rows of
// a chunk of CH steps are staged in LDS (the tables are in device memory by now: a chunk costs an L2 round trip).  sR holds r_t on entry.
template <int D, int CH>
__device__ __forceinline__ void head_backward(const KArgs<D>& ka, const double* __restrict__ sR, double* __restrict__ sTab /*[CH (D D + D)]*/, int lane, const double (&zeta)[D]) {
    constexpr int DD = D * D, ROW = DD + D;
    const double* __restrict__ tb = ka.tab;
    const int nhs = ka.nhs, n0 = ka.n0;
    const int row = lane < D ? lane : D - 1;
    double lam = 0.0, h[D];
#pragma unroll
    for (int k = 0; k < D; ++k) {
        lam = fma(tb[ka.to.Wm + row * D + k], zeta[k], lam);
        h[k] = tb[ka.to.h + k];
    }
    for (int hi = nhs; hi > 0; hi -= CH) {
        const int lo = hi - CH > 0 ? hi - CH : 0, cnt = hi - lo;
        lds_sync();
        {
            // the chunk's rows [G_t | c_t] are contiguous in the packed tables: 16-byte pieces, four loads in flight per lane and round
            // (a load-store-load chain would cost an L2 round trip per kilobyte; more in flight would cost the kernel's occupancy)
            const v2d* __restrict__ src = reinterpret_cast<const v2d*>(tb + ka.to.G + (size_t)lo * ROW);
            v2d* __restrict__ dst = reinterpret_cast<v2d*>(sTab);
            const int n2 = cnt * ROW / 2;
            for (int base = 0; base < n2; base += 4 * 64) {
                v2d b[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int idx = base + q * 64 + lane;
                    b[q] = src[idx < n2 ? idx : n2 - 1];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int idx = base + q * 64 + lane;
                    if (idx < n2) dst[idx] = b[q];
                }
            }
        }
        const int t = lo + lane, ti = t < n0 ? t : n0;
        const double cy = (lane < cnt) ? ka.y[t] : 0.0, crs = tb[ka.to.rS + ti], cr = (lane < cnt) ? sR[t] : 0.0;
        lds_sync();
        double mk = 0.0;
        // the row of the first step to be processed (the chunk's last)
        double g[D], cc;
        {
            const double* __restrict__ rw = sTab + (cnt - 1) * ROW;
#pragma unroll
            for (int k = 0; k < D; ++k) g[k] = rw[row * D + k];
            cc = rw[DD + row];
        }
        for (int s = cnt - 1; s >= 0; --s) {
            const int su = __builtin_amdgcn_readfirstlane(s);
            double gn[D], cn = 0.0;                  // the next step's row, in flight while this one's chain runs
            {
                const double* __restrict__ rw = sTab + (su > 0 ? su - 1 : 0) * ROW;
#pragma unroll
                for (int k = 0; k < D; ++k) gn[k] = rw[row * D + k];
                cn = rw[DD + row];
            }
            const double r = readlane_d(cr, su), yv = readlane_d(cy, su), rs = readlane_d(crs, su);
            double m = fma(-rs, r, yv), nl = cc * r;
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const double lk = readlane_d(lam, k);
                m = fma(h[k], lk, m);
                nl = fma(g[k], lk, nl);
            }
            lam = nl;
            mk = (lane == su) ? m : mk;
#pragma unroll
            for (int k = 0; k < D; ++k) g[k] = gn[k];
            cc = cn;
        }
        if (lane < cnt) ka.mean[t] = mk;
    }
}

// ---- the head by SCANS over its steps (d <= 4): a tile of 64 head steps at a time, lane t the affine map of step t -- (P_t, c_t) with
// P_t = M - db_t fw', c_t = (fb + db_t) u_t + fa forwards, (G_t, c_t r_t) backwards -- composed over the lanes in six levels of DPP moves
// (rows of 16 by row_shr / row_shl, then across the rows), d^2 + d values moved and d^3 + d^2 FMAs per level: ~400 instructions per
// 64 steps at d = 3 where the sequential form above spends 64 x 40 and a 64-step dependent chain (12.8 us against ~2).  From d = 5
// on an element no longer fits the registers the tiles' code leaves (3 (d^2 + d) doubles): those heads stay sequential.
// new <- mine o neighbour:  P <- P Pn,  c <- P cn + c   (row by row: a row of P is replaced once it has been used)
template <int D>
__device__ __forceinline__ void compose_after(double (&P)[D][D], double (&c)[D], const double (&Pn)[D][D], const double (&cn)[D]) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
        double row[D], ci = c[i];
#pragma unroll
        for (int k = 0; k < D; ++k) row[k] = 0.0;
#pragma unroll
        for (int m = 0; m < D; ++m) {
            ci = fma(P[i][m], cn[m], ci);
#pragma unroll
            for (int k = 0; k < D; ++k) row[k] = fma(P[i][m], Pn[m][k], row[k]);
        }
#pragma unroll
        for (int k = 0; k < D; ++k) P[i][k] = row[k];
        c[i] = ci;
    }
}
// the neighbour's element by one DPP control (lanes without a source, rows outside the mask: the identity), composed into mine
template <int D, int CTRL, int ROWMASK>
__device__ __forceinline__ void scan_level(double (&P)[D][D], double (&c)[D], bool has) {
    double Pn[D][D], cn[D];
    const double one = has ? 0.0 : 1.0;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        cn[i] = dpp_mov<CTRL, ROWMASK>(c[i]);
#pragma unroll
        for (int k = 0; k < D; ++k) Pn[i][k] = dpp_mov<CTRL, ROWMASK>(P[i][k]) + (i == k ? one : 0.0);
    }
    compose_after<D>(P, c, Pn, cn);
}

template <int D>
__device__ __forceinline__ void head_forward_scan(const KArgs<D>& ka, double* __restrict__ sR /*[kHeadMax]*/, int lane, double (&z0out)[D], double& quad) {
    const double* __restrict__ tb = ka.tab;
    const int nhs = ka.nhs, n0 = ka.n0;
    double z0[D];
#pragma unroll
    for (int i = 0; i < D; ++i) z0[i] = tb[ka.to.mu0 + i];      // (already in modal coordinates: V^-1 (A x0.m + a))
    double acc = 0.0;
    for (int t0 = 0; t0 < nhs; t0 += 64) {
        const int cnt = nhs - t0 < 64 ? nhs - t0 : 64;
        const bool valid = lane < cnt;
        const int t = valid ? t0 + lane : t0, ti = t < n0 ? t : n0;
        const double u = ka.y[t] - ka.hh, is = tb[ka.to.iS + ti];
        double P[D][D], c[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            const double db = tb[ka.to.db + ti * D + i];
            c[i] = valid ? fma(ka.fb[i] + db, u, ka.fa[i]) : 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const double m = (k == i) ? ka.fd[i] : ((k == partner<D>(i)) ? ka.fo[i] : 0.0);
                P[i][k] = valid ? fma(-db, ka.fw[k], m) : (i == k ? 1.0 : 0.0);
            }
        }
        const int p = lane & 15;
        scan_level<D, 0x111, 0xF>(P, c, p >= 1);
        scan_level<D, 0x112, 0xF>(P, c, p >= 2);
        scan_level<D, 0x114, 0xF>(P, c, p >= 4);
        scan_level<D, 0x118, 0xF>(P, c, p >= 8);
        scan_level<D, 0x142, 0xA>(P, c, (lane & 16) != 0);      // rows 1, 3: everything up to the end of the row below
        scan_level<D, 0x143, 0xC>(P, c, lane >= 32);            // rows 2, 3: everything up to lane 31
        // the state behind step t, in front of it (the left neighbour's; lane 0: the tile's start state), the innovation
        double za[D], r = u;
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double v = c[i];
#pragma unroll
            for (int k = 0; k < D; ++k) v = fma(P[i][k], z0[k], v);
            za[i] = v;
        }
#pragma unroll
        for (int i = 0; i < D; ++i) {
            const double sh = dpp_mov<0x138>(za[i]);
            r = fma(-ka.fw[i], lane == 0 ? z0[i] : sh, r);
        }
        if (valid) {
            acc = fma(r * r, is, acc);
            sR[t0 + lane] = r;
        }
#pragma unroll
        for (int i = 0; i < D; ++i) z0[i] = readlane_d(za[i], cnt - 1);
    }
    lds_sync();
#pragma unroll
    for (int i = 0; i < D; ++i) z0out[i] = z0[i];
    quad = readlane_d(wave_sum_to_lane63(acc), 63);
}

template <int D>
__device__ __forceinline__ void head_backward_scan(const KArgs<D>& ka, const double* __restrict__ sR, int lane, const double (&zeta)[D]) {
    constexpr int DD = D * D, ROW = DD + D;
    const double* __restrict__ tb = ka.tab;
    const int nhs = ka.nhs, n0 = ka.n0;
    double lamR[D], h[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) v = fma(tb[ka.to.Wm + i * D + k], zeta[k], v);
        lamR[i] = v;
        h[i] = tb[ka.to.h + i];
    }
    for (int t0 = ((nhs - 1) >> 6) << 6; t0 >= 0; t0 -= 64) {
        const int cnt = nhs - t0 < 64 ? nhs - t0 : 64;
        const bool valid = lane < cnt;
        const int t = valid ? t0 + lane : t0, ti = t < n0 ? t : n0;
        const double yt = ka.y[t], rs = tb[ka.to.rS + ti], r = sR[t];
        const double* __restrict__ rw = tb + ka.to.G + (size_t)t * ROW;      // [G_t | c_t], one row of the packed tables per head step
        double P[D][D], c[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            c[i] = valid ? rw[DD + i] * r : 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) P[i][k] = valid ? rw[i * D + k] : (i == k ? 1.0 : 0.0);
        }
        // suffix scan: lane t <- the steps t .. end of the tile (the neighbour holds the LATER steps, applied first)
        const int p = lane & 15;
        scan_level<D, 0x101, 0xF>(P, c, p + 1 < 16);
        scan_level<D, 0x102, 0xF>(P, c, p + 2 < 16);
        scan_level<D, 0x104, 0xF>(P, c, p + 4 < 16);
        scan_level<D, 0x108, 0xF>(P, c, p + 8 < 16);
        {
            // rows 0 and 2 take the first lane of the row above them: wave_shl:1 brings it to their lane 15, row_newbcast:15 spreads it
            double Pn[D][D], cn[D];
            const double one = (lane & 16) == 0 ? 0.0 : 1.0;
#pragma unroll
            for (int i = 0; i < D; ++i) {
                cn[i] = dpp_mov<0x15F, 0x5>(dpp_mov<0x130>(c[i]));
#pragma unroll
                for (int k = 0; k < D; ++k) Pn[i][k] = dpp_mov<0x15F, 0x5>(dpp_mov<0x130>(P[i][k])) + (i == k ? one : 0.0);
            }
            compose_after<D>(P, c, Pn, cn);
            // the lower half takes lane 32 (complete by now)
            const bool lower = lane < 32;
#pragma unroll
            for (int i = 0; i < D; ++i) {
                const double x = readlane_d(c[i], 32);
                cn[i] = lower ? x : 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    const double v = readlane_d(P[i][k], 32);
                    Pn[i][k] = lower ? v : (i == k ? 1.0 : 0.0);
                }
            }
            compose_after<D>(P, c, Pn, cn);
        }
        // lam behind step t (what the steps t .. end make of the tile's right-hand input); in front of it: the right neighbour's
        double lo[D], m = fma(-rs, r, yt);
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double v = c[i];
#pragma unroll
            for (int k = 0; k < D; ++k) v = fma(P[i][k], lamR[k], v);
            lo[i] = v;
        }
#pragma unroll
        for (int i = 0; i < D; ++i) {
            const double sh = dpp_mov<0x130>(lo[i]);
            m = fma(h[i], lane == 63 ? lamR[i] : sh, m);
        }
        if (valid) ka.mean[t0 + lane] = m;
#pragma unroll
        for (int i = 0; i < D; ++i) lamR[i] = readlane_d(lo[i], 0);
    }
}

// The head's variances: data-independent, the last stage of the tables (read in place from pinned memory: at most kHeadMax values)
template <int D>
__device__ __forceinline__ void head_variances(const KArgs<D>& ka, int lane) {
    wait_tables(ka.flag + 2, ka.seq);
    const double rn0 = ka.Rnew[0];
    const double* __restrict__ vb = ka.htab + ka.to.vb;
    for (int t = lane; t < ka.nhs; t += 64) ka.var[t] = vb[t < ka.n0 ? t : ka.n0] + (ka.rnew_per_step ? ka.RnewT[t] : rn0);
}

// =================================================================================================================================
// The kernel.  Workgroup g owns the steps [nhs + g C, nhs + (g + 1) C) (its core); its NW waves are NW consecutive tiles of 64 SUB steps
// that start `halo` steps earlier (g = 0: at nhs, with the head's exact end state) and end `halo` steps later.  Both mean recursions have
// forgotten a state after `halo` steps (tgp_steady_plan.hpp: |eigenvalue|^halo <= 2^-64), so a zero state at the start of the span and
// a zero lam at its end give every core step the same values, to rounding, as the recursion over the whole series: no pass over y
// before this one, no carries between workgroups.  Inside the workgroup the tiles are chained exactly (through LDS, over the at most
// three preceding / following tiles -- whatever lies further back has decayed as well).
// A lane holds SUB consecutive steps: 8, or 16 (the in-tile scans cost the same per level whatever a lane holds, so sixteen halve them
// per step -- at the price of 32 more registers).
// =================================================================================================================================
// The per-lane power table sPw[dir][re / im][component][lane]: job (dir, i) is one chain of 6 conditional complex multiplications by the bits of
// the lane's exponent; wave JOB mod NW runs it.  Direction and component are compile-time constants: every coefficient is a scalar load from
// the kernel arguments (a run-time index into the by-value argument struct would make the compiler keep a copy of it in scratch memory)
template <int D, int NW, int JOB>
struct PowerJobs {
    __device__ static __forceinline__ void run(const KArgs<D>& ka, double (*sPw)[2][D][64], int lane, int uw) {
        if constexpr (JOB < 2 * D) {
            if (uw == JOB % NW) {      // (wave-uniform)
                constexpr int dir = JOB / D, i = JOB % D;      // direction (0: forward, 1: backward), component
                const int e = dir == 0 ? lane : 63 - lane;
                double xr = 1.0, xi = 0.0;                      // the block form of M^(SUB e): (re, signed im) of the component, from the identity
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const double pr = dir == 0 ? ka.fpr[k][i] : ka.gpr[k][i], pi = dir == 0 ? ka.fpi[k][i] : ka.gpi[k][i];
                    const bool bit = ((e >> k) & 1) != 0;
                    // (a + i b)(c + i d) with the signed-imaginary convention: re = a c - b d, im = a d + b c (the sign of the pair's second member follows)
                    const double nr = fma(xr, pr, -(xi * pi)), ni = fma(xr, pi, xi * pr);
                    xr = bit ? nr : xr;
                    xi = bit ? ni : xi;
                }
                sPw[dir][0][i][lane] = xr;
                sPw[dir][1][i][lane] = xi;
            }
            PowerJobs<D, NW, JOB + 1>::run(ka, sPw, lane, uw);
        }
    }
};

// Development build (-DTGP_MODAL_PROBE): wave 0 of a mid-series workgroup stamps its phases (shader clock) into part[nwg + 12 ..]
#ifdef TGP_MODAL_PROBE
#define TGP_STAMP(k)                                                                     \
    do {                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                               \
        if (probe_wave) {                                                                \
            const double now = (double)clock64();                                        \
            if (lane == 0) ka.part[ka.nwg + 12 + (k)] = now;                             \
        }                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                               \
    } while (0)
#else
#define TGP_STAMP(k) do { } while (0)
#endif
// waves per SIMD the register allocation is held to (a workgroup of 8 waves puts two on each SIMD): d <= 2 fit four workgroups per CU as they
// are (64 registers), d = 3, 4 are held to three; beyond, what the code needs (d = 5 would spill at 80).  Measured with the half rows that
// freed the LDS for it: a fourth workgroup at d = 2 and a third at d = 4 change nothing -- in its steady state the kernel moves ~6 TB/s,
// what a copy reaches; its time at T = 1e7 is that plus the ramps at both ends
#define TGP_MIN_WAVES(D, NW, HS) ((NW) == 8 ? ((HS) ? 4 : (((D) == 3 || (D) == 4) ? 6 : 4)) : 2)
template <int D, int NW, int SUB, bool HEAD_SCANS>
__global__ __launch_bounds__(NW * 64, TGP_MIN_WAVES(D, NW, HEAD_SCANS)) void k_steady_one(const KArgs<D> ka_by_value) {
    // The arguments are read where they lie, in the kernel-argument segment (scalar loads, any index): a by-value struct is first copied to a
    // private variable, and one access pattern the optimiser cannot take apart (a run-time index, or identical blocks it merges into one
    // with the offsets in a phi) leaves the whole 3 KB struct in scratch memory -- 20 x the kernel's time, from one build to the next
    (void)ka_by_value;
    const KArgs<D>& ka = *(const KArgs<D>*)__builtin_amdgcn_kernarg_segment_ptr();      // (an address-space cast: C style)
    static_assert(SUB == kWJ, "eight steps per lane (sixteen were built and measured in round 4: the 32 more registers cost more occupancy than the halved scans save)");
    constexpr int TILE = 64 * SUB;
    constexpr int CHB = D <= 4 ? 64 : (D <= 6 ? 32 : 16);      // steps of the head's backward recursion staged in LDS at a time (~10 KB)
    __shared__ double sF[NW][D], sB[NW][D], sHead[D], sAcc[NW];
    __shared__ double sR[tgp_plan::kHeadMax];
    __shared__ __attribute__((aligned(16))) double sTab[(HEAD_SCANS && D <= 4) ? 2 : CHB * (D * D + D)];      // (the head as scans: nothing staged)
    __shared__ __attribute__((aligned(16))) double sOut[NW][2 * Row<SUB>::slots];
    // M^(SUB l) (forward) and Mg^(SUB (63 - l)) (backward) for the lanes l of a tile: what carries a tile's start state / right-hand input to
    // its lanes.  The same for every wave: the waves share the building (by the bits of the lane number) between them
    __shared__ double sPw[2][2][D][64];
    __shared__ double sPoison;      // NaN once a bounded wait for the host ran out (host head): the call's result is then discarded
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) sPoison = 0.0;
    // XCD-aware order: consecutive workgroups (which share their halos' lines of y) land on the same XCD, hence the same L2
    long long wg;
    {
        const long long per = (ka.nwg + 7) / 8;
        wg = (long long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
        if ((long long)(blockIdx.x >> 3) >= per || wg >= ka.nwg) return;
    }
    const long long T = ka.T, Tv = ka.T_eff;      // Tv: the steps this launch may read (a segment: its own and `halo` beyond)
    const long long g = ka.wg0 + wg;               // the workgroup's number in the series
    long long c_lo = ka.nhs + g * ka.C, c_hi_raw = c_lo + ka.C;
    c_lo = c_lo > ka.seg_lo ? c_lo : ka.seg_lo;
    c_hi_raw = c_hi_raw < ka.seg_hi ? c_hi_raw : ka.seg_hi;
    const long long c_hi = c_hi_raw < T ? c_hi_raw : T;
    const bool first = g == 0 && ka.seg_lo == 0;      // the workgroup behind the head (a later segment may begin inside the series' first core range)
    const long long s0 = first ? (long long)ka.nhs : c_lo - ka.halo;
    const long long tile_t0 = s0 + (long long)wave * TILE, t0 = tile_t0 + (long long)lane * SUB;
    const bool head_wave = first && wave == 0;
    const bool any_valid = tile_t0 < Tv;                             // (wave-uniform)
    const bool need_back = any_valid && tile_t0 + TILE > c_lo;       // a tile wholly inside the left halo only hands its end state on
    const bool has_out = ka.post && need_back && tile_t0 < c_hi;

    const bool probe_wave = wg == (ka.nwg >> 1) && wave == 0;
    (void)probe_wave;
    const bool probe = wg == (ka.nwg >> 1) && threadIdx.x == 0;      // (TGP_STEADY_DEBUG: a mid-series workgroup's lifetime in shader cycles and in 100 MHz ticks)
    if (probe) {
        ka.part[ka.nwg + 8] = (double)clock64();
        ka.part[ka.nwg + 9] = (double)wall_clock64();
    }
    double head_quad = 0.0;
    if (first && ka.hosthead) {
        if (wave == NW - 1) {      // the head's inputs to the host, first thing: its forward recursion runs there
            for (int t = lane; t < ka.nhs; t += 64) {
                ka.head_in[t] = ka.y[t];
                if (ka.post && ka.rnew_per_step) ka.head_in[ka.nhs + t] = ka.RnewT[t];
            }
            if (ka.post && !ka.rnew_per_step && lane == 0) ka.head_in[ka.nhs] = ka.Rnew[0];
            __threadfence_system();
            if (lane == 0) __hip_atomic_store(ka.hflag, 2 * ka.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    } else if (first) {
        // the head's tables: wait for the host, pull them into device memory (all waves), then the head wave runs the head forward
        if (threadIdx.x == 0) ka.part[ka.nwg + 1] = (double)wall_clock64();      // (phases of workgroup 0, 100 MHz: TGP_STEADY_DEBUG prints them)
        wait_tables(ka.flag, ka.seq, threadIdx.x == 0 ? &ka.part[ka.nwg] : nullptr);      // (timed out: the head's share becomes NaN)
        if (threadIdx.x == 0) ka.part[ka.nwg + 2] = (double)wall_clock64();
        const v2d* __restrict__ src = reinterpret_cast<const v2d*>(ka.htab);
        v2d* __restrict__ dst = reinterpret_cast<v2d*>(ka.tab);
        for (int idx = threadIdx.x; idx < ka.to.G / 2; idx += NW * 64) dst[idx] = src[idx];      // (stage 0: everything in front of the rows [G_t | c_t])
        __syncthreads();
        if (head_wave) {
            double z0[D];
            __builtin_amdgcn_s_setprio(3);      // (the one sequential wave of the launch: ahead of the tiles' waves it shares its SIMD with)
            if constexpr (HEAD_SCANS && D <= 4) head_forward_scan<D>(ka, sR, lane, z0, head_quad);
            else head_forward<D>(ka, sR, lane, z0, head_quad);
            __builtin_amdgcn_s_setprio(0);
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < D; ++i) sHead[i] = z0[i];
                ka.part[ka.nwg + 3] = (double)wall_clock64();
            }
        } else if (ka.post) {
            // meanwhile the other waves fetch stage 1, the rows of the head's backward recursion (the barrier below hands them to the head wave)
            wait_tables(ka.flag + 1, ka.seq);
            for (int idx = ka.to.G / 2 + (int)threadIdx.x - 64; idx < ka.to.vb / 2; idx += (NW - 1) * 64) dst[idx] = src[idx];
        }
    }
    // ---- forward, zero start: innovations r0 of the lane's steps, the lane's end state, inclusive scan over the lanes
    double yv[SUB], r[SUB], st[D];
    const long long left = Tv - t0;
    const int nvalid = left >= SUB ? SUB : (left > 0 ? (int)left : 0);
    // (2 D chains of 6 conditional complex multiplications -- shared out over the waves, one component of one direction each, while the
    //  observations are on their way; two waves doing all of it made the other six wait at the barrier: 9 000 cycles at d = 6)
    auto build_powers = [&]() { PowerJobs<D, NW, 0>::run(ka, sPw, lane, __builtin_amdgcn_readfirstlane(wave)); };
    TGP_STAMP(0);
    if (any_valid) load_lane<SUB, D>(ka, t0, yv);
    // (the new noise is wanted at the very end, by the variance stores: fetched here, beside the observations, not in front of those stores)
    const double rn0 = (ka.post && !ka.rnew_per_step) ? ka.Rnew[0] : 0.0;
    build_powers();
    __syncthreads();      // (the in-tile scans below read the table; the observations are still on their way)
    {
        double z[D];
#pragma unroll
        for (int i = 0; i < D; ++i) z[i] = 0.0;
        if (any_valid) {
#ifdef TGP_MODAL_PROBE
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            TGP_STAMP(1);
#pragma unroll
            for (int j = 0; j < SUB; ++j) {
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
            TGP_STAMP(2);
            // inclusive scan over the lanes without an LDS round trip: four levels inside the rows of 16 lanes (row_shr moves), then the rows'
            // end states across (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3) through the per-lane powers of the table
            TGP_ROW_LEVEL(0, 0, fpr, fpi, z);
            TGP_ROW_LEVEL(0, 1, fpr, fpi, z);
            TGP_ROW_LEVEL(0, 2, fpr, fpi, z);
            TGP_ROW_LEVEL(0, 3, fpr, fpi, z);
            {
                const int e1 = (lane & 15) + 1;                 // lane 16 r + p takes M^(SUB (p + 1)) times the state behind lane 16 r - 1
                double g[D];
#pragma unroll
                for (int i = 0; i < D; ++i) g[i] = dpp_mov<0x142, 0xA>(z[i]);
#pragma unroll
                for (int i = 0; i < D; ++i) z[i] = fma(sPw[0][0][i][e1], g[i], fma(sPw[0][1][i][e1], g[partner<D>(i)], z[i]));
                const int e2 = lane >= 32 ? lane - 31 : 0;      // lane l >= 32 takes M^(SUB (l - 31)) times the state behind lane 31
#pragma unroll
                for (int i = 0; i < D; ++i) g[i] = dpp_mov<0x143, 0xC>(z[i]);
#pragma unroll
                for (int i = 0; i < D; ++i) z[i] = fma(sPw[0][0][i][e2], g[i], fma(sPw[0][1][i][e2], g[partner<D>(i)], z[i]));
            }
        }
#pragma unroll
        for (int i = 0; i < D; ++i) st[i] = dpp_mov<0x138>(z[i]);      // the state in front of the lane's steps: its left neighbour's (lane 0: zero)
        if (lane == 63) {
#pragma unroll
            for (int i = 0; i < D; ++i) sF[wave][i] = any_valid ? z[i] : 0.0;
        }
    }
    TGP_STAMP(3);
#ifdef TGP_MODAL_PROBE
    if (wg == (ka.nwg >> 1) && lane == 0) ka.part[ka.nwg + 28 + wave] = (double)clock64();      // every wave's arrival at barrier 1 ...
#endif
    __syncthreads();
    TGP_STAMP(4);
    double acc = 0.0;
    double zst[D];
#pragma unroll
    for (int i = 0; i < D; ++i) zst[i] = 0.0;
    if (any_valid) {
        // the tile's start state from the (at most three) tiles before it; workgroup 0: the head's end state sits at position -1
        double zin[D];
#pragma unroll
        for (int i = 0; i < D; ++i) zin[i] = 0.0;
        if (first && ka.hosthead && wave < 3) wait_tables(ka.hflag + 1, ka.seq, &sPoison);      // (wave-uniform) the head's end state comes from the host: z0p below
#pragma unroll
        for (int k = 1; k <= 3; ++k) {
            const int src = wave - k;
            if (src < -1 || (src == -1 && !first)) continue;      // (wave-uniform)
            double x[D];
#pragma unroll
            for (int i = 0; i < D; ++i) x[i] = (src >= 0) ? sF[src][i] : (ka.hosthead ? ka.z0p[i] : sHead[i]);
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
        // the lane's start state: st + M^(SUB lane) zin
#pragma unroll
        for (int i = 0; i < D; ++i) st[i] = fma(sPw[0][0][i][lane], zin[i], fma(sPw[0][1][i][lane], zin[partner<D>(i)], st[i]));
        const bool in_core = t0 >= c_lo && t0 < c_hi_raw;
        // the start state moves step j's innovation by -fw' M^j st
#pragma unroll
        for (int j = 0; j < SUB; ++j) {
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
        acc = wave_sum_to_lane63(acc);
        if (lane == 63) sAcc[wave] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) t += sAcc[w];
            ka.part[wg] = t + sPoison;
            if (first) ka.part[ka.nwg] = head_quad;
        }
        return;
    }
    // ---- backward, zero lam behind the tile: m0_j = y_j - rS r_j + gw . zeta (zeta from the lane's own later steps), reverse scan
    TGP_STAMP(5);
    if (need_back) {
        double zeta[D];
#pragma unroll
        for (int i = 0; i < D; ++i) zeta[i] = 0.0;
#pragma unroll
        for (int j = SUB - 1; j >= 0; --j) {
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
        TGP_STAMP(6);
        // the reverse scan: rows by row_shl moves; then rows 0 and 2 take the first lane of the row above them (wave_shl:1 brings it to their
        // lane 15, row_newbcast:15 spreads it), then the lower half takes lane 32 (a scalar by v_readlane)
        TGP_ROW_LEVEL(1, 0, gpr, gpi, zeta);
        TGP_ROW_LEVEL(1, 1, gpr, gpi, zeta);
        TGP_ROW_LEVEL(1, 2, gpr, gpi, zeta);
        TGP_ROW_LEVEL(1, 3, gpr, gpi, zeta);
        {
            const int e1 = 47 + (lane & 15);                  // Mg^(SUB (16 - p)) sits at position 63 - (16 - p) of the backward table
            double g[D];
#pragma unroll
            for (int i = 0; i < D; ++i) g[i] = dpp_mov<0x15F, 0x5>(dpp_mov<0x130>(zeta[i]));
#pragma unroll
            for (int i = 0; i < D; ++i) zeta[i] = fma(sPw[1][0][i][e1], g[i], fma(sPw[1][1][i][e1], g[partner<D>(i)], zeta[i]));
            const int e2 = lane < 32 ? 31 + lane : 63;        // Mg^(SUB (32 - l)) at position 63 - (32 - l); the upper half: the identity's place, times zero
            const double keep = lane < 32 ? 1.0 : 0.0;
#pragma unroll
            for (int i = 0; i < D; ++i) g[i] = readlane_d(zeta[i], 32) * keep;
#pragma unroll
            for (int i = 0; i < D; ++i) zeta[i] = fma(sPw[1][0][i][e2], g[i], fma(sPw[1][1][i][e2], g[partner<D>(i)], zeta[i]));
        }
#pragma unroll
        for (int i = 0; i < D; ++i) zst[i] = dpp_mov<0x130>(zeta[i]);      // the lam behind the lane's steps: its right neighbour's (lane 63: zero)
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < D; ++i) sB[wave][i] = zeta[i];
        }
    } else if (lane == 0) {
#pragma unroll
        for (int i = 0; i < D; ++i) sB[wave][i] = 0.0;
    }
    TGP_STAMP(7);
#ifdef TGP_MODAL_PROBE
    if (wg == (ka.nwg >> 1) && lane == 0) ka.part[ka.nwg + 28 + NW + wave] = (double)clock64();      // ... and at barrier 2
#endif
    __syncthreads();
    TGP_STAMP(8);
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
#pragma unroll
        for (int i = 0; i < D; ++i) zst[i] = fma(sPw[1][0][i][lane], zin[i], fma(sPw[1][1][i][lane], zin[partner<D>(i)], zst[i]));
        v2d* row = reinterpret_cast<v2d*>(sOut[wave]);
        const int wb = Row<SUB>::lane_slot(lane & 31);
        // the lam behind the lane's last step reaches step j through Mg^(SUB - 1 - j): WG holds gw' Mg^(7 - j)
#pragma unroll
        for (int j = 0; j < SUB; ++j) {
            double m = yv[j];
#pragma unroll
            for (int i = 0; i < D; ++i) m = fma(ka.WG[j][i], zst[i], m);
            yv[j] = m;
        }
        TGP_STAMP(9);
        // out through the wave's LDS row, half a tile at a time (the lower 32 lanes' values, then the upper 32's)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if ((lane >> 5) == half) {
#pragma unroll
                for (int j = 0; j < SUB; j += 2) {
                    v2d w;
                    w.x = yv[j];
                    w.y = yv[j + 1];
                    row[wb + (j >> 1)] = w;
                }
            }
            lds_sync();
            flush_row<SUB>(ka.mean, tile_t0 + half * (TILE / 2), c_lo, c_hi, row, lane);
            lds_sync();
        }
        TGP_STAMP(10);
        // the variances do not depend on the data: a constant outside the last n1 steps (plus the new noise); written transposed as well
        {
            long long n1 = 0;
            if (T - tile_t0 <= (long long)tgp_plan::kTailMax + TILE) {      // (wave-uniform: the last few tiles of the series)
                wait_tables(ka.flag + 2, ka.seq);
                n1 = (long long)ka.htab[0];
            }
            const double* __restrict__ tvb = ka.htab + ka.to.tvb;      // (in place from pinned memory: the last few tiles, a handful of values)
            const bool aligned = (reinterpret_cast<uintptr_t>(ka.var) & 15) == 0 && (!ka.rnew_per_step || (reinterpret_cast<uintptr_t>(ka.RnewT) & 15) == 0);
            // (wave-uniform: a tile inside the workgroup's range, away from the series' end, one shared new noise -- nearly every tile: one constant)
            const bool plain = aligned && !ka.rnew_per_step && n1 == 0 && tile_t0 >= c_lo && tile_t0 + TILE <= c_hi;
            if (plain) {
                v2d w;
                w.x = ka.vb + rn0;
                w.y = w.x;
                v2d* q = reinterpret_cast<v2d*>(ka.var + tile_t0);
#pragma unroll
                for (int k = 0; k < SUB / 2; ++k) store_pair(q + k * 64 + lane, w);
            } else
#pragma unroll
            for (int k = 0; k < SUB / 2; ++k) {
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
                        rn = *reinterpret_cast<const v2d*>(ka.RnewT + t);
                    } else {
                        rn.x = rn0;
                        rn.y = rn0;
                    }
                    v2d w;
                    w.x = v0 + rn.x;
                    w.y = v1 + rn.y;
                    store_pair(reinterpret_cast<v2d*>(ka.var + t), w);
                } else {
                    if (t >= c_lo && t < c_hi) ka.var[t] = v0 + (ka.rnew_per_step ? ka.RnewT[t] : rn0);
                    if (t + 1 >= c_lo && t + 1 < c_hi) ka.var[t + 1] = v1 + (ka.rnew_per_step ? ka.RnewT[t + 1] : rn0);
                }
            }
        }
    }
    TGP_STAMP(11);
    if (head_wave && ka.hosthead) {
        double zin[D];
        right_input(-1, zin);
        if (lane == 0) {      // the backward state entering the head: the host runs the head backwards from it
#pragma unroll
            for (int i = 0; i < D; ++i) ka.zeta_out[i] = zin[i];
            __threadfence_system();
            __hip_atomic_store(ka.hflag + 2, 2 * ka.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    } else if (head_wave) {
        double zin[D];
        right_input(-1, zin);
        if (lane == 0) ka.part[ka.nwg + 4] = (double)wall_clock64();
        __builtin_amdgcn_s_setprio(3);
        if constexpr (HEAD_SCANS && D <= 4) head_backward_scan<D>(ka, sR, lane, zin);
        else head_backward<D, CHB>(ka, sR, sTab, lane, zin);
        __builtin_amdgcn_s_setprio(0);
        if (lane == 0) ka.part[ka.nwg + 5] = (double)wall_clock64();
        head_variances<D>(ka, lane);
    }
    // the head's outputs come from the host: a workgroup from the MIDDLE of the dispatch writes them -- the host has them tens of microseconds after
    // workgroup 0 has handed zeta back, so it does not wait, and its read over PCIe is not the kernel's tail (the last workgroup doing it was
    // measured: + 4 us on the kernel)
    if (ka.hosthead && ka.head_out != nullptr && wg == ((ka.nwg + 7) / 8) / 2 && wave < 2) {
        wait_tables(ka.hflag + 3, ka.seq, &sPoison);
        double* dst = wave == 0 ? ka.mean : ka.var;
        const double* src = ka.head_out + (wave == 0 ? 0 : ka.nhs);
        for (int t = lane; t < ka.nhs; t += 64) dst[t] = src[t];
    }
    acc = wave_sum_to_lane63(acc);
    if (lane == 63) sAcc[wave] = acc;
    TGP_STAMP(12);
    __syncthreads();
    TGP_STAMP(13);
    if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += sAcc[w];
        ka.part[wg] = t + sPoison;
        if (first) ka.part[ka.nwg] = head_quad;
        if (wg == ka.nwg - 1) ka.part[ka.nwg + 6] = (double)wall_clock64();
        if (probe) {
            ka.part[ka.nwg + 10] = (double)clock64();
            ka.part[ka.nwg + 11] = (double)wall_clock64();
        }
    }
}

// block-form helpers of the host side (the sign convention of the imaginary parts is preserved by squaring: re^2 - im^2, 2 re im)
static void bsquare(int d, const double* ar, const double* ai, double* outr, double* outi) {
    for (int i = 0; i < d; ++i) {
        const double r = ar[i] * ar[i] - ai[i] * ai[i], im = 2.0 * ar[i] * ai[i];
        outr[i] = r;
        outi[i] = im;
    }
}

template <int D>
void fill_args(KArgs<D>& ka, const Modal& md, int sub) {
    for (int i = 0; i < D; ++i) {
        ka.fd[i] = md.fd[i]; ka.fo[i] = md.fo[i]; ka.fb[i] = md.fb[i]; ka.fa[i] = md.fa[i]; ka.fw[i] = md.fw[i];
        ka.gd[i] = md.gd[i]; ka.go[i] = md.go[i]; ka.gc[i] = md.gc[i]; ka.gw[i] = md.gw[i];
        for (int j = 0; j < kWJ; ++j) {
            ka.WJ[j][i] = md.WJ[j][i];
            ka.WG[j][i] = md.WG[j][i];
        }
    }
    // in-tile scan levels M^(sub 2^k)
    if (sub == 8) {
        std::memcpy(ka.fpr[0], md.fp8r, sizeof(double) * D); std::memcpy(ka.fpi[0], md.fp8i, sizeof(double) * D);
        std::memcpy(ka.gpr[0], md.gp8r, sizeof(double) * D); std::memcpy(ka.gpi[0], md.gp8i, sizeof(double) * D);
    } else {
        bsquare(D, md.fp8r, md.fp8i, ka.fpr[0], ka.fpi[0]);
        bsquare(D, md.gp8r, md.gp8i, ka.gpr[0], ka.gpi[0]);
    }
    for (int k = 1; k < 6; ++k) {
        bsquare(D, ka.fpr[k - 1], ka.fpi[k - 1], ka.fpr[k], ka.fpi[k]);
        bsquare(D, ka.gpr[k - 1], ka.gpi[k - 1], ka.gpr[k], ka.gpi[k]);
    }
    // tile powers M^TILE, M^(2 TILE)
    if (sub == 8) {
        std::memcpy(ka.ftr[0], md.fp512r, sizeof(double) * D); std::memcpy(ka.fti[0], md.fp512i, sizeof(double) * D);
        std::memcpy(ka.gtr[0], md.gp512r, sizeof(double) * D); std::memcpy(ka.gti[0], md.gp512i, sizeof(double) * D);
    } else {
        bsquare(D, md.fp512r, md.fp512i, ka.ftr[0], ka.fti[0]);
        bsquare(D, md.gp512r, md.gp512i, ka.gtr[0], ka.gti[0]);
    }
    bsquare(D, ka.ftr[0], ka.fti[0], ka.ftr[1], ka.fti[1]);
    bsquare(D, ka.gtr[0], ka.gti[0], ka.gtr[1], ka.gti[1]);
    ka.hh = md.hh; ka.rS = md.rS; ka.vb = md.vb;
    ka.n0 = md.n0; ka.nhs = md.nhs; ka.halo = md.halo;
}

}  // namespace

struct Engine {
    HeadTables* tab = nullptr;      // host scratch of the plan
    double* hflat = nullptr;        // pinned host memory: the packed tables of the call (TabOff) + the flag word behind them
    double* dflat = nullptr;        // device memory: where workgroup 0 puts them
    size_t flat_cap = 0;            // doubles
    TabOff to{};
    size_t flat_used = 0;
    bool post = true;               // the launched call wants the posterior marginals (complete(): which stages of the tables)
    double* part = nullptr;         // pinned host memory
    size_t part_cap = 0;
    Modal md{};
    tgp_plan::Info info{};
    long long nwg = 0, seq = 0;      // workgroups of the whole series
    long long nwg_local = 0, wg0 = 0;      // ... of the last launch (a time segment runs a slice of them)
    bool owns_head = true;
    int nw = 8, sub = 8;
    bool began = false, deferred = false;
    // the head on the host (hosthead): pinned [0, 2 kHeadMax) head inputs y | Rnew, [2, 4 kHeadMax) head outputs mean | var, then z0 [8], zeta [8],
    // four flag words; hr: the head's innovations (host scratch)
    double* hhead = nullptr;
    double hr[tgp_plan::kHeadMax];
    bool hosthead = false, hh_pending = false;
    int rnew_per_step = 0;
    hipStream_t stream = nullptr;      // of the last launch (the host's waits watch it)
    double host_quad = 0.0;
    tgp_plan::ModelHost mh{};
    long long hh_T = 0;
    // the streaming logpdf kernel (tgp_lml.hip): a logpdf-only call over the whole series
    bool lml = false;
    tgp_lml::Geometry lg{};
    double lWt[tgp_plan::kMaxD * tgp_plan::kMaxD];
    bool lml_records = false;          // the host may take the launched kernel's end from its records in pinned memory (tgp_lml.hpp records_there)
    void* pxch = nullptr;              // device memory: the exchange records of the streaming posterior kernel's runs (tgp_post.hpp)
    long long stream_min_T = -1;       // TGP_OPT_STREAM_MIN_T
    // the core of the last plan, kept while the model and the length stand (the reference's own sequence -- logpdf(model, y), then posterior(model, y) --
    // plans the same model twice): key = the model's blocks as handed over
    std::vector<double> memo_key;
    long long memo_T = -1;
    bool memo_ok = false;
    unsigned long long memo_stamp = 0; // of the thread's plan workspace when the core was built (another engine's plan on this thread, another thread: no hit)
    Modal memo_md{};
    tgp_plan::Info memo_info{};
    long long memo_Wt_n = -1;          // lWt holds quad_table(md, memo_Wt_n)
};
namespace {
constexpr size_t kHH = tgp_plan::kHeadMax;
inline double* hh_in(Engine* e) { return e->hhead; }
inline double* hh_out(Engine* e) { return e->hhead + 2 * kHH; }
inline double* hh_z0(Engine* e) { return e->hhead + 4 * kHH; }
inline double* hh_zeta(Engine* e) { return e->hhead + 4 * kHH + 8; }
inline long long* hh_flag(Engine* e) { return reinterpret_cast<long long*>(e->hhead + 4 * kHH + 16); }
}  // namespace

Engine* create() { return new Engine(); }
void set_stream_min_T(Engine* e, long long v) {
    if (e) e->stream_min_T = v;
}
namespace {
// The streaming kernels' crossovers against k_steady_one (scripts/r06_kernel_T.py, d = 3: fixed 18 / 24 us + 0.7 / 2.9 us per 1e6 steps against
// 12 / 22 us + 1.9 / 3.8): below them the short-lived workgroups of k_steady_one finish first
constexpr long long kLmlMinT = 5000000, kPostMinT = 3000000;
inline bool stream_serves(const Engine* e, long long T, long long crossover) { return T >= (e->stream_min_T < 0 ? crossover : e->stream_min_T); }
}  // namespace
void destroy(Engine* e) {
    if (!e) return;
    delete e->tab;
    if (e->hflat) (void)tgp_alloc::host_free(e->hflat);
    if (e->hhead) (void)tgp_alloc::host_free(e->hhead);
    if (e->dflat) (void)tgp_alloc::dev_free(e->dflat);
    if (e->part) (void)tgp_alloc::host_free(e->part);
    if (e->pxch) (void)tgp_alloc::dev_free(e->pxch);
    delete e;
}

const tgp_plan::Info& last_plan(const Engine* e) { return e->info; }
const char* kernel_name(const Engine* e, bool post) {
    if (e->lml && !post) return e->lg.n == 32 ? "k_lml_stream<32>" : "k_lml_stream<16>";
    return e->nw == 8 ? (post ? "k_steady_one<8x8,posterior>" : "k_steady_one<8x8,logpdf>") : (post ? "k_steady_one<16x8,posterior>" : "k_steady_one<16x8,logpdf>");
}
const tgp_plan::Modal& last_modal(const Engine* e) { return e->md; }

namespace {
static bool head_scans_enabled() {      // TGP_MODAL_HEAD_SCANS=0: the sequential head everywhere (A/B runs)
    static const bool on = [] {
        const char* v = std::getenv("TGP_MODAL_HEAD_SCANS");
        return !(v && v[0] == '0');
    }();
    return on;
}

// The streaming posterior kernel (tgp_post.hip) serves a call over the WHOLE series with the head on the host and every series-sized
// buffer on a 16-byte boundary; segments, in-kernel heads and odd pointers stay on k_steady_one.
static bool use_post_stream(const Engine* e, const Call& c) {
    if (c.mean == nullptr || !e->hosthead || c.seg_lo != 0 || (c.seg_hi >= 0 && c.seg_hi < c.T)) return false;
    if (!tgp_post::applies(e->md, c.T) || !stream_serves(e, c.T, kPostMinT)) return false;
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return al(c.y) && al(c.mean) && al(c.var) && (!c.rnew_per_step || al(c.Rnew));
}

template <int D>
int launch(Engine* e, hipStream_t st, const Call& c, const char** kname) {
    if (use_post_stream(e, c)) {
        if (!e->pxch) {
            if (tgp_alloc::dev_malloc(&e->pxch, tgp_post::xch_bytes()) != hipSuccess) return (int)hipErrorOutOfMemory;
            if (hipMemsetAsync(e->pxch, 0, tgp_post::xch_bytes(), st) != hipSuccess) return (int)hipGetLastError();
        }
        const tgp_post::Geometry g = tgp_post::choose_geometry(e->md, c.T);
        tgp_post::Call pc;
        pc.T = c.T;
        pc.y = c.y;
        pc.Rnew = c.Rnew;
        pc.rnew_per_step = c.rnew_per_step;
        pc.mean = c.mean;
        pc.var = c.var;
        pc.htab = e->hflat;
        pc.tvb_off = e->to.tvb;
        pc.flag = reinterpret_cast<const long long*>(e->hflat + e->flat_cap);
        pc.part = e->part;
        pc.head_in = hh_in(e);
        pc.z0p = hh_z0(e);
        pc.zeta_out = hh_zeta(e);
        pc.head_out = hh_out(e);
        pc.hflag = hh_flag(e);
        pc.seq = e->seq;
        pc.xch = e->pxch;
        e->post = true;
        e->rnew_per_step = c.rnew_per_step;
        e->stream = st;
        e->wg0 = 0;
        e->nwg_local = g.nwg;
        e->owns_head = true;
        e->hh_pending = true;
        return tgp_post::enqueue(st, e->md, g, pc, kname);
    }
    KArgs<D> ka;
    static_assert(sizeof(KArgs<D>) <= 4096, "the kernel-argument segment");
    std::memset(&ka, 0, sizeof ka);
    fill_args<D>(ka, e->md, e->sub);
    ka.post = c.mean != nullptr ? 1 : 0;
    e->post = ka.post != 0;
    ka.rnew_per_step = c.rnew_per_step;
    ka.T = c.T;
    const int nw = e->nw, tile = 64 * e->sub;
    ka.C = (long long)nw * tile - 2LL * e->md.halo;
    // the launch's slice of the series' workgroups: all of them, or those that own outputs of the segment [seg_lo, seg_hi)
    const long long seg_lo = c.seg_lo, seg_hi = (c.seg_hi < 0 || c.seg_hi > c.T) ? c.T : c.seg_hi;
    const long long nhs = e->md.nhs;
    const long long g0 = seg_lo <= nhs ? 0 : (seg_lo - nhs) / ka.C, g1 = (seg_hi - nhs + ka.C - 1) / ka.C;
    e->wg0 = g0;
    e->nwg_local = g1 - g0;
    e->owns_head = seg_lo == 0;
    ka.nwg = e->nwg_local;
    ka.wg0 = g0;
    ka.seg_lo = seg_lo;
    ka.seg_hi = seg_hi;
    ka.T_eff = (seg_hi + e->md.halo < c.T) ? seg_hi + e->md.halo : c.T;
    ka.yl = c.yl;
    ka.yr = c.yr;
    ka.seq = e->seq;
    ka.y = c.y - seg_lo;
    ka.Rnew = c.Rnew;
    ka.RnewT = (c.Rnew && c.rnew_per_step) ? c.Rnew - seg_lo : c.Rnew;
    ka.mean = c.mean ? c.mean - seg_lo : nullptr;
    ka.var = c.var ? c.var - seg_lo : nullptr;
    ka.htab = e->hflat;
    ka.tab = e->dflat;
    ka.flag = reinterpret_cast<const long long*>(e->hflat + e->flat_cap);
    ka.to = e->to;
    ka.part = e->part;
    e->hh_pending = false;
    e->rnew_per_step = c.rnew_per_step;
    e->stream = st;
    if (e->hosthead && e->owns_head) {
        ka.hosthead = 1;
        ka.head_in = hh_in(e);
        ka.z0p = hh_z0(e);
        ka.zeta_out = hh_zeta(e);
        ka.head_out = ka.post ? hh_out(e) : nullptr;
        ka.hflag = hh_flag(e);
        e->hh_pending = true;
    }
    const long long per = (e->nwg_local + 7) / 8;
    const unsigned grid = (unsigned)(per * 8);
    *kname = kernel_name(e, ka.post != 0);
    // the head as scans (d <= 4) needs ~30 more registers per lane than the tiles' code: taken where the launch leaves the CUs room anyway
    // (at most two workgroups per CU: every series up to ~2e6 steps -- there the head IS the call), not where occupancy is the kernel's time
    const bool head_scans = D <= 4 && e->owns_head && e->nwg_local <= 512 && head_scans_enabled() && !ka.hosthead;
    if (nw == 8) {
        if (head_scans) hipLaunchKernelGGL((k_steady_one<D, 8, 8, (D <= 4)>), dim3(grid), dim3(8 * 64), 0, st, ka);
        else hipLaunchKernelGGL((k_steady_one<D, 8, 8, false>), dim3(grid), dim3(8 * 64), 0, st, ka);
    } else {
        hipLaunchKernelGGL((k_steady_one<D, 16, 8, false>), dim3(grid), dim3(16 * 64), 0, st, ka);
    }
    return (int)hipGetLastError();
}
}  // namespace

const char* kernel_name(const Engine* e, const Call& c) {
    if (use_post_stream(e, c)) return "k_post_stream";
    return kernel_name(e, c.mean != nullptr);
}

// TGP_MODAL_GEOMETRY=<waves>x<steps per lane> (8x8, 16x8) overrides the choice (A/B runs)
void choose_geometry(int d, int halo, int* nw, int* sub) {
    static const int forced = [] {
        const char* v = std::getenv("TGP_MODAL_GEOMETRY");
        int a = 0, b = 0;
        if (v && std::sscanf(v, "%dx%d", &a, &b) == 2 && b == 8 && (a == 8 || a == 16)) return a * 100 + b;
        return 0;
    }();
    const bool wide = 2 * halo * 10 > 3 * 4096;      // halos beyond ~30 % of a 4096-step span: spans of 8192 steps
    *sub = 8;
    *nw = wide ? 16 : 8;
    (void)d;
    if (forced) {
        *nw = forced / 100;
        *sub = forced % 100;
        if (halo * 2 >= *nw * 64 * *sub) {      // (a forced geometry too small for the halo: the wide default)
            *sub = 8;
            *nw = 16;
        }
    }
}

// This translation unit's host code is compiled for AVX2 + FMA (Makefile: the plan's small matrix products vectorise 4-wide -- an 8 x 8
// product 36 ns instead of 150); a host without them gets the five-launch engine.
static bool host_cpu_ok() {
    static const bool ok = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
    return ok;
}

// The host plans other translation units need (tgp_api.hip: the filter / posterior / rand one-launch paths).  Defined HERE and only here: this
// object's host code is built with -mavx2 -mfma, and the inline functions of tgp_steady_plan.hpp instantiated in a second object built with
// other flags would be two definitions of one entity -- the linker keeps either (round-4 advice).  All behind the run-time CPU check: a
// host without AVX2 gets a plan that declines (the older engines serve the call).
void plan_filter(const tgp_plan::ModelHost& m, long long T, tgp_plan::FilterPlan& fp) {
    if (!host_cpu_ok()) {
        fp.why = tgp_plan::kEigFail;
        return;
    }
    tgp_plan::build_filter_any(m, T, fp);
}
void plan_filter_head(const tgp_plan::ModelHost& m, const tgp_plan::FilterPlan& fp, const double* y, double* mout, double* Pout, double* mu_end, double* quad) {
    tgp_plan::filter_head_any(m, fp, y, mout, Pout, mu_end, quad);      // (only ever called with a plan plan_filter accepted)
}
bool plan_posterior_head(const tgp_plan::ModelHost& m, const tgp_plan::FilterPlan& fp, const double* y, double* Gh, double* gh, double* Lh, double* Gss,
                         double* Lss, double* mu_end, double* quad) {
    return tgp_plan::posterior_head_any(m, fp, y, Gh, gh, Lh, Gss, Lss, mu_end, quad);
}
void plan_rand(const tgp_plan::ModelHost& m, tgp_plan::RandPlan& rp) {
    if (!host_cpu_ok()) {
        rp.why = tgp_plan::kEigFail;
        return;
    }
    tgp_plan::build_rand_any(m, rp);
}

tgp_plan::Info plan_only(const tgp_plan::ModelHost& m, long long T, tgp_plan::Modal& md, tgp_plan::HeadTables& tab) {
    if (!host_cpu_ok()) {
        tgp_plan::Info in;
        in.why = tgp_plan::kEigFail;
        return in;
    }
    return tgp_plan::build_any(m, T, md, tab);
}

// TGP_MODAL_OVERLAP=0: the whole plan before the launch (A/B runs).  Also when the process runs with synchronous launches (the usual ROCm /
// PyTorch debugging switches): the kernel waits for flags the host raises AFTER hipLaunchKernelGGL returns -- a launch that does not return
// until the kernel has finished would wait for itself (round-4 advice; the wait is bounded as well, see wait_tables).
static bool hosthead_enabled() {      // TGP_MODAL_HOSTHEAD=0: the head inside the kernel as in round 4 (A/B runs)
    static const bool on = [] {
        const char* v = std::getenv("TGP_MODAL_HOSTHEAD");
        return !(v && v[0] == '0');
    }();
    return on;
}
static bool overlap_tables() {
    static const bool on = [] {
        const char* v = std::getenv("TGP_MODAL_OVERLAP");
        if (v && v[0] == '0') return false;
        for (const char* name : {"HIP_LAUNCH_BLOCKING", "CUDA_LAUNCH_BLOCKING", "AMD_SERIALIZE_KERNEL", "AMD_SERIALIZE_COPY"}) {
            const char* b = std::getenv(name);
            if (b && b[0] != '\0' && !(b[0] == '0' && b[1] == '\0')) return false;
        }
        return true;
    }();
    return on;
}

// the layout of the packed tables (fixed by d and n0: known when the kernel is launched, before the tables themselves)
static void layout_tables(Engine* e) {
    const int d = e->md.d, dd = d * d, n = e->md.n0 + 1;
    TabOff& to = e->to;
    int off = 1;
    auto take = [&](int cnt) {
        const int o = off;
        off += cnt;
        return o;
    };
    to.h = take(d);
    to.mu0 = take(d);
    to.Wm = take(dd);
    to.db = take(n * d);
    to.iS = take(n);
    to.rS = take(n);
    off += off & 1;                                   // (16-byte pieces)
    to.G = take(e->md.nhs * (dd + d));                // rows [G_t | c_t], one per head step t < nhs (steps behind n0: the stationary row)
    to.c = to.G;
    to.vb = take(n);
    to.tvb = take(0);
}

// packs one stage of the plan's tables for the device (pinned memory) and raises the stage's flag: 0 what the head's forward recursion
// reads, 1 the rows [G_t | c_t] of its backward recursion, 2 the smoothed variances of the head and of the last n1 steps.  A stage that
// declined (why != kOk) raises its flag and the later ones with the failure bit: the kernel never waits for a stage that will not come
static void ship_stage(Engine* e, int stage, int why) {
    const Modal& md = e->md;
    const HeadTables& tb = *e->tab;
    const int d = md.d, dd = d * d, n = md.n0 + 1;
    const TabOff& to = e->to;
    double* q = e->hflat;
    long long* hflag = reinterpret_cast<long long*>(e->hflat + e->flat_cap);
    if (why != tgp_plan::kOk) {
        for (int s2 = stage; s2 < 3; ++s2) __atomic_store_n(hflag + s2, 2 * e->seq + 1, __ATOMIC_RELEASE);
        return;
    }
    if (stage == 0) {
        std::memcpy(q + to.h, tb.h, sizeof(double) * d);
        std::memcpy(q + to.mu0, tb.mu0, sizeof(double) * d);
        std::memcpy(q + to.Wm, tb.Wm, sizeof(double) * dd);
        std::memcpy(q + to.db, tb.kA, sizeof(double) * n * d);
        std::memcpy(q + to.iS, tb.iS, sizeof(double) * n);
        std::memcpy(q + to.rS, tb.rS, sizeof(double) * n);
    } else if (stage == 1) {
        for (int t = 0; t < md.nhs; ++t) {
            const int tt = t < md.n0 ? t : md.n0;
            std::memcpy(q + to.G + (size_t)t * (dd + d), tb.G + (size_t)tt * dd, sizeof(double) * dd);
            std::memcpy(q + to.G + (size_t)t * (dd + d) + dd, tb.c + (size_t)tt * d, sizeof(double) * d);
        }
    } else {
        std::memcpy(q + to.vb, tb.vb, sizeof(double) * n);
        const int n1 = md.n1 > 0 ? md.n1 : 0;
        std::memcpy(q + to.tvb, tb.tvb, sizeof(double) * n1);
        q[0] = (double)n1;
        e->flat_used = (size_t)to.tvb + n1;
    }
    __atomic_store_n(hflag + stage, 2 * e->seq, __ATOMIC_RELEASE);
}

// the tables half of the plan, stage by stage, each shipped as soon as it exists.  Returns the Why of the first stage that declined
static int build_and_ship_tables(Engine* e, long long T, int nstages = 3) {
    e->memo_ok = false;      // (stage 0 rewrites the head's gains in place, in modal coordinates: the core in `tab` is no longer what build_core left)
    for (int stage = 0; stage < nstages; ++stage) {
        const int why = tgp_plan::build_tables_stage_any(e->md.d, stage, T, e->md, *e->tab, e->info);
        ship_stage(e, stage, why);
        if (why != tgp_plan::kOk) return why;
    }
    return tgp_plan::kOk;
}

static bool lml_stream_enabled() {      // TGP_LML_STREAM=0: logpdf on k_steady_one as in round 5 (A/B runs)
    static const bool on = [] {
        const char* v = std::getenv("TGP_LML_STREAM");
        return !(v && v[0] == '0');
    }();
    return on;
}

static bool lml_records_enabled() {      // TGP_LML_RECORDS=0: the end of a logpdf-only launch by hipStreamSynchronize as before (A/B runs)
    static const bool on = [] {
        const char* s = std::getenv("TGP_LML_RECORDS");
        return !(s && s[0] == '0');
    }();
    return on;
}
namespace {
bool plan_memo_enabled() {      // TGP_PLAN_MEMO=0: every call plans from scratch (A/B runs)
    static const bool on = [] {
        const char* s = std::getenv("TGP_PLAN_MEMO");
        return !(s && s[0] == '0');
    }();
    return on;
}
template <class F>
void model_words(const tgp_plan::ModelHost& m, F&& f) {
    const size_t d = (size_t)m.d;
    f(m.A, d * d); f(m.a, d); f(m.Q, d * d); f(m.H, d); f(m.hh, 1); f(m.R, 1); f(m.x0m, d); f(m.x0P, d * d);
}
bool memo_hit(const Engine* e, const tgp_plan::ModelHost& m, long long T) {
    if (!e->memo_ok || !plan_memo_enabled() || e->memo_T != T || e->memo_md.d != m.d) return false;
    if (tgp_plan::work_stamp_any(m.d) != e->memo_stamp) return false;      // (the stages behind the core read the thread's workspace: it must still be this plan's)
    const double* k = e->memo_key.data();
    bool same = true;
    model_words(m, [&](const double* p, size_t n) {
        same = same && std::memcmp(k, p, n * sizeof(double)) == 0;
        k += n;
    });
    return same;
}
void memo_store(Engine* e, const tgp_plan::ModelHost& m, long long T) {
    e->memo_key.clear();
    model_words(m, [&](const double* p, size_t n) { e->memo_key.insert(e->memo_key.end(), p, p + n); });
    e->memo_T = T;
    e->memo_md = e->md;
    e->memo_info = e->info;
    e->memo_stamp = tgp_plan::work_stamp_any(m.d);
    e->memo_ok = true;
}
}  // namespace
bool plan(Engine* e, const tgp_plan::ModelHost& m, long long T, bool logpdf_only) {
    e->began = false;
    e->lml = false;
    if (!host_cpu_ok()) {
        e->info = tgp_plan::Info{};
        e->info.why = tgp_plan::kEigFail;
        return false;
    }
    if (!e->tab) {
        e->tab = new HeadTables();
        // the layout of ship_tables at its largest: d = 8, n0 = kN0Max, n1 = kTailMax
        e->flat_cap = (size_t)tgp_plan::kHeadMax * (64 + 8) + (size_t)(tgp_plan::kN0Max + 1) * (8 + 3) + tgp_plan::kTailMax + 64 + 2 * 8 + 8;
        e->flat_cap = (e->flat_cap + 1) & ~(size_t)1;
        if (tgp_alloc::host_malloc(reinterpret_cast<void**>(&e->hflat), (e->flat_cap + 4) * sizeof(double), hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess ||
            tgp_alloc::dev_malloc(reinterpret_cast<void**>(&e->dflat), (e->flat_cap + 4) * sizeof(double)) != hipSuccess) {
            e->info = tgp_plan::Info{};
            e->info.why = tgp_plan::kEigFail;
            return false;
        }
        for (int s2 = 0; s2 < 3; ++s2) reinterpret_cast<long long*>(e->hflat + e->flat_cap)[s2] = 0;      // (the stages' flags)
    }
    if (!e->hhead) {
        const size_t n = 4 * kHH + 16 + 4;
        if (tgp_alloc::host_malloc(reinterpret_cast<void**>(&e->hhead), n * sizeof(double), hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) {
            e->info = tgp_plan::Info{};
            e->info.why = tgp_plan::kEigFail;
            return false;
        }
        std::memset(e->hhead, 0, n * sizeof(double));
    }
    ++e->seq;
    if (memo_hit(e, m, T)) {
        e->md = e->memo_md;
        e->info = e->memo_info;
    } else {
        e->memo_ok = false;
        e->memo_Wt_n = -1;
        e->info = tgp_plan::build_core_any(m, T, e->md, *e->tab);
        if (e->info.why != tgp_plan::kOk) return false;
        memo_store(e, m, T);
    }
    layout_tables(e);
    e->mh = m;
    e->hh_T = T;
    if (logpdf_only && lml_stream_enabled() && stream_serves(e, T, kLmlMinT)) {
        // the streaming logpdf kernel: no tables (the head runs on the host from build_core's gains), no wait inside the kernel
        e->lg = tgp_lml::choose_geometry(e->md, T);
        const size_t need = tgp_lml::part_doubles(e->md.d) + 64;      // (the records and the development stamps of TGP_LML_DBG)
        if (need > e->part_cap) {
            if (e->part) (void)tgp_alloc::host_free(e->part);
            e->part = nullptr;
            e->part_cap = 0;
            if (tgp_alloc::host_malloc(reinterpret_cast<void**>(&e->part), need * sizeof(double), hipHostMallocDefault) != hipSuccess) {
                e->info.why = tgp_plan::kEigFail;
                return false;
            }
            std::memset(e->part, 0, need * sizeof(double));
            e->part_cap = need;
        }
        if (e->memo_Wt_n != e->lg.first_tile) {
            tgp_lml::quad_table(e->md, e->lg.first_tile, e->lWt);
            e->memo_Wt_n = e->memo_ok ? e->lg.first_tile : -1;
        }
        e->deferred = false;
        e->hosthead = false;
        e->lml = true;
        e->nwg = e->lg.nwg;
        e->began = true;
        return true;
    }
    // the tables half: behind the launch when the series is certainly longer than head + tail (the kernel's head wave and last tiles wait
    // for the flag), else right here
    // (measured again in round 5, d = 3: with the tables behind the launch a call takes 63.5 us against 68.7 with them in front of it, the
    //  kernel 47 us either way: scripts/call_overhead.py; TGP_MODAL_OVERLAP=0 is the switch)
    e->deferred = overlap_tables() && T >= (long long)e->md.nhs + tgp_plan::kTailMax + 1;
    // the head on the host wherever the tables may follow the launch: workgroup 0 waits for the head's end state, not for tables
    e->hosthead = e->deferred && hosthead_enabled();
    if (!e->deferred) {
        const int why = build_and_ship_tables(e, T);
        if (why != tgp_plan::kOk) {
            e->info.why = why;
            return false;
        }
    }
    // geometry: steps per lane (8 or 16) and tiles per workgroup; spans of 4096 steps unless the halos would eat more than ~30 % of them, then 8192
    const int halo = e->md.halo;
    choose_geometry(e->md.d, halo, &e->nw, &e->sub);
    const long long C = (long long)e->nw * 64 * e->sub - 2LL * halo;
    e->nwg = (T - e->md.nhs + C - 1) / C;
    const size_t need = std::max<size_t>((size_t)e->nwg + 64, 8192);      // (room for the streaming kernels' triples and development stamps)
    if (need > e->part_cap) {
        if (e->part) (void)tgp_alloc::host_free(e->part);
        e->part = nullptr;
        e->part_cap = 0;
        if (tgp_alloc::host_malloc(reinterpret_cast<void**>(&e->part), need * sizeof(double), hipHostMallocDefault) != hipSuccess) {
            e->info.why = tgp_plan::kEigFail;
            return false;
        }
        e->part_cap = need;
    }
    e->began = true;
    return true;
}

// Behind the launch: the tables half of the plan, if it was deferred.  false: it declined (the kernel has been released all the same -- whatever
// it writes is to be discarded, the caller re-runs the call elsewhere after synchronising).
namespace {
}  // namespace
// The host's side of a hand-over: spins on a flag in pinned memory the kernel raises.  Not bounded by the clock -- the kernel may not even
// have started (a stream with work of the caller's in front of it) -- but by the stream: once it has drained (hipStreamQuery, every few
// thousand spins) the kernel is through, and the flag is there or will never be (the kernel's own waits give up after two seconds and
// poison its result).  false: the stream is done, or in error, and the flag never came.
bool await_host_flag(const long long* f, long long v, hipStream_t stream) {
    for (unsigned spin = 1;; ++spin) {
        if (__atomic_load_n(f, __ATOMIC_ACQUIRE) >= v) return true;
        if ((spin & 4095u) == 0) {
            const hipError_t q = hipStreamQuery(stream);
            (void)hipGetLastError();      // (hipErrorNotReady is not an error of ours: it must not surface in the next launch's check)
            if (q != hipErrorNotReady) return __atomic_load_n(f, __ATOMIC_ACQUIRE) >= v;
        }
        __builtin_ia32_pause();
    }
}
namespace {
void raise_flag(long long* f, long long v) {
    if (__atomic_load_n(f, __ATOMIC_RELAXED) < v) __atomic_store_n(f, v, __ATOMIC_RELEASE);
}
}  // namespace

bool complete(Engine* e, long long T) {
    if (e->began && e->lml) {
        // the head's forward recursion, as soon as the kernel has handed its observations over (or the stream has drained)
        // (false: the stream has drained without the flag -- a short series, whose head wave raises none: the kernel's end has delivered the data.
        //  A launch that failed is the caller's stream synchronisation's to report; the head of garbage is then discarded with the rest)
        (void)await_host_flag(hh_flag(e), 2 * e->seq, e->stream);
        tgp_plan::modal_head_forward_any(e->mh, e->md, *e->tab, hh_in(e), e->hr, hh_z0(e), &e->host_quad);
        return true;
    }
    if (!e->began || !e->deferred) return true;
    e->deferred = false;
    if (!e->hosthead) {
        const int why = build_and_ship_tables(e, T, e->post ? 3 : 1);      // (a logpdf-only launch reads the forward stage alone)
        if (why != tgp_plan::kOk) {
            e->info.why = why;
            return false;
        }
        return true;
    }
    // ---- the head on the host: its forward recursion as soon as workgroup 0 has handed the observations over, the tables of stages 1 and 2
    // beside the kernel (stage 2 is shipped: the last tiles read the tail variances), the backward recursion once the kernel has handed zeta back
    long long* f = hh_flag(e);
    const long long v = 2 * e->seq;
    const bool head = e->hh_pending;
    e->hh_pending = false;
    bool ok = true;
    int why = tgp_plan::kOk;
    // the tail variances FIRST (round 6): the runs / workgroups at the series' end wait for them from the moment they start, and they need nothing of
    // the head (measured on k_post_stream: shipped behind the head's tables they left the series' last run waiting until 50 us into a 35 us kernel)
    if (e->post) {
        why = tgp_plan::build_tables_tail_any(e->md.d, T, e->md, *e->tab, e->info);
        ship_stage(e, 2, why);      // (the head's variances in that stage are not there yet: with the head on the host nobody on the device reads them)
    }
    if (head) {
        ok = await_host_flag(f, v, e->stream);
        if (ok) tgp_plan::modal_head_forward_any(e->mh, e->md, *e->tab, hh_in(e), e->hr, hh_z0(e), &e->host_quad);
        raise_flag(f + 1, v);
    }
    if (e->post && why == tgp_plan::kOk) {
        why = tgp_plan::build_tables_stage_any(e->md.d, 1, T, e->md, *e->tab, e->info);
        if (why == tgp_plan::kOk) why = tgp_plan::build_tables_headvar_any(e->md.d, e->md, *e->tab);
    }
    if (head && e->post) {
        if (ok && why == tgp_plan::kOk) ok = await_host_flag(f + 2, v, e->stream);
        if (ok && why == tgp_plan::kOk) {
            const int nhs = e->md.nhs;
            double *om = hh_out(e), *ov = om + nhs;
            const double* in = hh_in(e);
            tgp_plan::modal_head_backward_any(e->md, *e->tab, in, e->hr, hh_zeta(e), om, ov);
            for (int t = 0; t < nhs; ++t) ov[t] += in[nhs + (e->rnew_per_step ? t : 0)];
        }
        raise_flag(f + 3, v);      // (also when something declined: the last workgroup never waits for what will not come; the outputs are discarded)
    }
    if (why != tgp_plan::kOk) {
        e->info.why = why;
        return false;
    }
    if (!ok) {
        e->info.why = tgp_plan::kEigFail;      // (the host / device hand-over timed out)
        return false;
    }
    return true;
}

// The kernel of a logpdf-only call produces nothing but the workgroups' records in pinned memory, each pair of them marked with the call's key: once
// every record is there the kernel has read all of y and written all it writes, and the host goes on without hipStreamSynchronize (the kernel's own
// end -- the release at the end of the dispatch, the queue's barrier packet, the completion signal -- is ~4 us the caller need not wait for; the
// stream stays ordered: whatever is enqueued next runs behind the kernel).  false: the stream drained or failed without them -- synchronise.
bool await_done(Engine* e) {
    if (!e || !e->began || !e->lml || !e->lml_records) return false;
    size_t next = 0;
    for (unsigned spin = 1;; ++spin) {
        if (tgp_lml::records_there(e->lg, e->md.d, e->part, e->seq, &next)) return true;
        if ((spin & 4095u) == 0) {
            const hipError_t q = hipStreamQuery(e->stream);
            (void)hipGetLastError();
            if (q != hipErrorNotReady) return tgp_lml::records_there(e->lg, e->md.d, e->part, e->seq, &next);
        }
        __builtin_ia32_pause();
    }
}

// A caller that leaves between enqueue() and complete() (an error return in between) must not leave the kernel waiting: raises the flags of
// a deferred plan with the failure bit (what the kernel writes is then discarded by whoever reads it).  Harmless otherwise.
void abandon(Engine* e) {
    if (!e || !e->began || !e->deferred || !e->hflat) return;
    e->deferred = false;
    long long* hflag = reinterpret_cast<long long*>(e->hflat + e->flat_cap);
    for (int s2 = 0; s2 < 3; ++s2) __atomic_store_n(hflag + s2, 2 * e->seq + 1, __ATOMIC_RELEASE);
    if (e->hhead) {
        raise_flag(hh_flag(e) + 1, 2 * e->seq);
        raise_flag(hh_flag(e) + 3, 2 * e->seq);
    }
    e->hh_pending = false;
}

int enqueue(Engine* e, hipStream_t stream, const Call& c, const char** kname, std::string* err) {
    if (!e || !e->began || !c.y || c.T <= 0 || (c.mean && (!c.var || !c.Rnew))) {
        if (err) *err = "tgp_modal::enqueue: bad argument / no plan";
        return (int)hipErrorInvalidValue;
    }
    int r = 0;
    if (e->lml) {
        if (c.mean != nullptr || c.seg_lo != 0 || (c.seg_hi >= 0 && c.seg_hi < c.T)) {
            if (err) *err = "tgp_modal::enqueue: the plan was made for a logpdf-only call over the whole series";
            return (int)hipErrorInvalidValue;
        }
        tgp_lml::Buffers b;
        b.part = e->part;
        b.head_in = hh_in(e);
        b.flags = hh_flag(e);
        e->lml_records = lml_records_enabled();
        e->post = false;
        e->stream = stream;
        e->owns_head = true;
        e->nwg_local = e->lg.nwg;
        r = tgp_lml::enqueue(stream, e->md, e->lg, c.T, c.y, b, e->seq, e->lWt, kname);
        if (r != 0 && err) *err = std::string("tgp_lml: launch: ") + hipGetErrorString((hipError_t)r);
        return r;
    }
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

// After the stream has passed the kernel: the launch's share of the quadratic form -- the sum of r^2 over the steps it owns behind the head
// and (the launch that holds the head) the head's sum of r^2 / S_t; fixed summation order.
void finish_parts(const Engine* e, double* ssq, double* head_quad) {
    if (e->lml) {
        *ssq = tgp_lml::finish(e->md, e->lg, e->part, e->hhead + 4 * kHH, e->lWt);
        *head_quad = e->host_quad;
        return;
    }
    double s = 0.0;
    for (long long g = 0; g < e->nwg_local; ++g) s += e->part[g];
    *ssq = s;
    *head_quad = e->owns_head ? (e->hosthead ? e->host_quad : e->part[e->nwg_local]) : 0.0;
}

// ... and the log marginal likelihood of a call that ran the whole series.
double finish(const Engine* e, long long T) {
    const Modal& md = e->md;
    if (std::getenv("TGP_POST_DBG") != nullptr && (std::atoi(std::getenv("TGP_POST_DBG")) & 16) && e->post && e->nwg_local <= 256) {
        // development: the runs' start / end stamps of k_post_stream (100 MHz ticks)
        const double* q = e->part + 512;
        const long long R = e->nwg_local * 8;
        double s0 = 1e300, s1 = -1e300, e0 = 1e300, e1 = -1e300, esum = 0.0;
        long long elast = -1, n = 0;
        for (long long r = 0; r < R; ++r) {
            if (q[2 * r] == 0.0) continue;
            ++n;
            s0 = std::min(s0, q[2 * r]); s1 = std::max(s1, q[2 * r]);
            e0 = std::min(e0, q[2 * r + 1]);
            esum += q[2 * r + 1];
            if (q[2 * r + 1] > e1) { e1 = q[2 * r + 1]; elast = r; }
        }
        {
            // histogram of the ends (2 us bins), mean end by XCD (workgroup number mod 8) and by wave slot
            int hist[64] = {0};
            double xs[8] = {0}, ws[8] = {0};
            int xn[8] = {0}, wn[8] = {0};
            for (long long r = 0; r < R; ++r) {
                if (q[2 * r] == 0.0) continue;
                const double en = (q[2 * r + 1] - s0) * 0.01;
                const int b = (int)(en / 2.0);
                ++hist[b < 63 ? b : 63];
                xs[(r / 8) % 8] += en; ++xn[(r / 8) % 8];
                ws[r % 8] += en; ++wn[r % 8];
            }
            fprintf(stderr, "[tgp post] ends by 2 us:");
            for (int b = 0; b < 40; ++b) if (hist[b]) fprintf(stderr, " %d:%d", 2 * b, hist[b]);
            fprintf(stderr, "\n[tgp post] mean end by XCD:");
            for (int x = 0; x < 8; ++x) fprintf(stderr, " %.1f", xs[x] / std::max(1, xn[x]));
            fprintf(stderr, "; by wave of the workgroup:");
            for (int x = 0; x < 8; ++x) fprintf(stderr, " %.1f", ws[x] / std::max(1, wn[x]));
            fprintf(stderr, "\n");
        }
        fprintf(stderr, "[tgp post] %lld runs: starts within %.2f us; ends: first %.2f, mean %.2f, last %.2f us after the first start (run %lld); run 0 ends %.2f, run R/2 ends %.2f\n", n,
                (s1 - s0) * 0.01, (e0 - s0) * 0.01, (esum / n - s0) * 0.01, (e1 - s0) * 0.01, elast, (q[1] - s0) * 0.01, (q[2 * (R / 2) + 1] - s0) * 0.01);
    }
    if (e->lml && std::getenv("TGP_LML_DBG") != nullptr) {
        const double* q = e->part + tgp_lml::kStampOff;
        double s0 = 1e300, s1 = 0, l1 = 0, c1 = 0, b1 = 0, e1 = 0;
        for (int g = 0; g < e->lg.nwg && g < 512; ++g) {
            s0 = std::min(s0, q[8 * g]);
            s1 = std::max(s1, q[8 * g]);
        }
        for (int g = 0; g < e->lg.nwg && g < 512; ++g) {
            l1 = std::max(l1, q[8 * g + 1] - s0); c1 = std::max(c1, q[8 * g + 2] - s0); b1 = std::max(b1, q[8 * g + 3] - s0); e1 = std::max(e1, q[8 * g + 4] - s0);
        }
        fprintf(stderr, "[tgp lml] %d workgroups; wave 0 of each, us after the first start: starts within %.2f; first tile landed by %.2f; runs done by %.2f; barrier passed by %.2f; triples written by %.2f\n",
                e->lg.nwg, (s1 - s0) * 0.01, l1 * 0.01, c1 * 0.01, b1 * 0.01, e1 * 0.01);
    }
    if (!e->lml && std::getenv("TGP_STEADY_DEBUG") != nullptr) {
        const double* q = e->part + e->nwg_local;
        fprintf(stderr, "[tgp modal] d %d n0 %d nhs %d n1 %d halo %d geometry %dx%d workgroups %lld | workgroup 0 (us from its start): tables ready %.1f, head forward done %.1f, head backward starts %.1f, done %.1f; last workgroup done %.1f\n",
                md.d, md.n0, md.nhs, md.n1, md.halo, e->nw, e->sub, e->nwg_local, (q[2] - q[1]) * 0.01, (q[3] - q[1]) * 0.01, (q[4] - q[1]) * 0.01, (q[5] - q[1]) * 0.01, (q[6] - q[1]) * 0.01);
        if (q[11] > q[9])
            fprintf(stderr, "[tgp modal] a mid-series workgroup lived %.2f us = %.0f shader cycles (%.0f MHz); it started %.1f us after workgroup 0\n", (q[11] - q[9]) * 0.01, q[10] - q[8],
                    (q[10] - q[8]) / ((q[11] - q[9]) * 0.01), (q[9] - q[1]) * 0.01);
#ifdef TGP_MODAL_PROBE
        {
            static const char* nm[14] = {"entry", "y loaded", "forward steps", "forward scan", "barrier 1", "corrections", "backward steps", "backward scan", "barrier 2", "mean rows in LDS",
                                         "mean stores issued", "variance stores issued", "wave sum", "barrier 3"};
            fprintf(stderr, "[tgp modal] phases of its wave 0 (shader cycles since entry):");
            for (int k = 1; k < 14; ++k) fprintf(stderr, " %s %.0f;", nm[k], q[12 + k] - q[12]);
            fprintf(stderr, "\n[tgp modal] its waves reached barrier 1 at");
            for (int w = 0; w < e->nw; ++w) fprintf(stderr, " %.0f", q[28 + w] - q[12]);
            fprintf(stderr, "; barrier 2 at");
            for (int w = 0; w < e->nw; ++w) fprintf(stderr, " %.0f", q[28 + e->nw + w] - q[12]);
            fprintf(stderr, "\n");
        }
#endif
    }
    double s = 0.0, hq = 0.0;
    finish_parts(e, &s, &hq);
    const double quad = hq + md.iS * s;
    const double logdet = md.LS + (double)(T - md.n0) * md.logS;
    return -0.5 * ((double)T * kLog2Pi + logdet + quad);
}

// =================================================================================================================================
// rand of an LTI model (lgssm.jl:65-91, the draws supplied): x_t = A x_{t-1} + a + Lq eps_t,  y_t = h' x_t + hh + sqrt(R) eta_t.
// The same one-launch structure as k_steady_one, forwards only and on DENSE powers (the open-loop transition of a Matern-3/2 / -5/2 block is
// a Jordan block: no modal form).  Workgroup g owns the outputs [g C, (g + 1) C), C = 8 tiles of 512 steps minus the halo; its tiles start
// `halo` steps earlier from a zero state (g = 0: at step 0 from the drawn x0); a lane runs its 8 steps from zero, the lanes' end states are
// scanned (rows of 16 by DPP moves with A^(8 2^k), across the rows through a per-lane table of A^(8 e) in LDS), the tiles are chained over
// the <= 3 tiles before them, and the lane's start state reaches its 8 outputs through the rows h' A^(j+1).  Reads eps_t (8 d B/step) and
// eps_e once, writes y once.
// =================================================================================================================================
namespace {
template <int D>
struct RArgs {
    double A[D][D], a[D], Lq[D][D], h[D], hh, sR;
    double P[6][D][D];       // A^(8 2^k)
    double PT[2][D][D];      // A^512, A^1024
    double WJ[kWJ][D];       // h' A^(j+1)
    double x0[D];
    long long T, C, nwg;
    int halo;
    const double *eps_t, *eps_e;
    double* y;
};

// y <- M x (dense, M from the kernel arguments)
#define TGP_MATVEC_ACC(M, X, Y)                                   \
    _Pragma("unroll") for (int i_ = 0; i_ < D; ++i_) {            \
        double v_ = Y[i_];                                        \
        _Pragma("unroll") for (int k_ = 0; k_ < D; ++k_) v_ = fma(M[i_][k_], X[k_], v_); \
        Y[i_] = v_;                                               \
    }

template <int D, int NW>
__global__ __launch_bounds__(NW * 64, (D <= 4 ? 4 : 2)) void k_rand_one(const RArgs<D> by_value) {
    (void)by_value;
    const RArgs<D>& ka = *(const RArgs<D>*)__builtin_amdgcn_kernarg_segment_ptr();      // (read in place: see k_steady_one)
    constexpr int SUB = kWJ, TILE = 64 * SUB;
    __shared__ double sF[NW][D];
    __shared__ double sPw[D][D][64];      // A^(8 e), e = 0 .. 63: entry [i][k][e]
    constexpr int LV = SUB * D / 2;       // 16-byte pieces of draws per lane
    __shared__ v2d sT[NW][16 * (LV + 1)];      // a quarter of a tile's draws on their way from memory order to the lanes (one spare piece per lane:
                                               // the 16 reading lanes then start in 16 different groups of four banks)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long g;
    {
        const long long per = (ka.nwg + 7) / 8;      // consecutive workgroups (they share their halos' lines) on the same XCD
        g = (long long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
        if ((long long)(blockIdx.x >> 3) >= per || g >= ka.nwg) return;
    }
    const long long T = ka.T, c_lo = g * ka.C, c_hi = (c_lo + ka.C < T) ? c_lo + ka.C : T;
    const bool first = g == 0;
    const long long s0 = first ? 0 : c_lo - ka.halo;
    const long long tile_t0 = s0 + (long long)wave * TILE, t0 = tile_t0 + (long long)lane * SUB;
    const bool any_valid = tile_t0 < c_hi;      // (wave-uniform: a tile wholly behind the workgroup's range has nothing to do)
    // ---- the per-lane power table: row i of A^(8 e) by the bits of e, rows shared out over the waves
    {
        const int uw = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
        for (int i = 0; i < D; ++i) {
            if (uw != i % NW) continue;      // (wave-uniform)
            double row[D];
#pragma unroll
            for (int k = 0; k < D; ++k) row[k] = (k == i) ? 1.0 : 0.0;
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                double nr[D];
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    double v = 0.0;
#pragma unroll
                    for (int m = 0; m < D; ++m) v = fma(row[m], ka.P[b][m][k], v);
                    nr[k] = v;
                }
                const bool bit = ((lane >> b) & 1) != 0;
#pragma unroll
                for (int k = 0; k < D; ++k) row[k] = bit ? nr[k] : row[k];
            }
#pragma unroll
            for (int k = 0; k < D; ++k) sPw[i][k][lane] = row[k];
        }
    }
    // ---- the lane's eight steps from a zero state
    double y0[SUB], x[D];
#pragma unroll
    for (int i = 0; i < D; ++i) x[i] = 0.0;
#pragma unroll
    for (int j = 0; j < SUB; ++j) y0[j] = 0.0;
    if (any_valid) {
        const bool whole = t0 + SUB <= T && (reinterpret_cast<uintptr_t>(ka.eps_t) & 15) == 0 && (reinterpret_cast<uintptr_t>(ka.eps_e) & 15) == 0;
        // a whole tile's draws are 64 SUB D consecutive doubles: fetched in memory order (every load instruction 1 KB of consecutive bytes;
        // a lane fetching its own SUB D values, 16 bytes at a stride of 8 SUB D, keeps the address path busy four to eight times as long)
        // and handed to their lanes through LDS, sixteen lanes' worth at a time
        const bool tile_whole = tile_t0 + TILE <= T && (reinterpret_cast<uintptr_t>(ka.eps_t) & 15) == 0 && ((tile_t0 * D) & 1) == 0;      // (wave-uniform)
        double evt[SUB * D];
        if (tile_whole) {
            const v2d* src = reinterpret_cast<const v2d*>(ka.eps_t + tile_t0 * D);
            v2d ld[4][D];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < D; ++i) ld[r][i] = src[r * 16 * LV + i * 64 + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    const int idx = i * 64 + lane;
                    sT[wave][idx + idx / LV] = ld[r][i];
                }
                lds_sync();
                if ((lane >> 4) == r) {
#pragma unroll
                    for (int k = 0; k < LV; ++k) {
                        const v2d w = sT[wave][(lane & 15) * (LV + 1) + k];
                        evt[2 * k] = w.x;
                        evt[2 * k + 1] = w.y;
                    }
                }
                lds_sync();
            }
        }
        double ee[SUB];
        if (whole) {
            const v2d* q = reinterpret_cast<const v2d*>(ka.eps_e + t0);
#pragma unroll
            for (int j = 0; j < SUB / 2; ++j) {
                const v2d w = q[j];
                ee[2 * j] = w.x;
                ee[2 * j + 1] = w.y;
            }
        } else {
#pragma unroll
            for (int j = 0; j < SUB; ++j) ee[j] = t0 + j < T ? ka.eps_e[t0 + j] : 0.0;
        }
#pragma unroll
        for (int j2 = 0; j2 < SUB; j2 += 2) {
            // the draws of two steps: 2 d consecutive doubles (16-byte pieces where the series allows)
            double ev[2 * D];
            if (tile_whole) {
#pragma unroll
                for (int k = 0; k < 2 * D; ++k) ev[k] = evt[j2 * D + k];
            } else if (whole) {
                const v2d* q = reinterpret_cast<const v2d*>(ka.eps_t + (t0 + j2) * D);
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    const v2d w = q[k];
                    ev[2 * k] = w.x;
                    ev[2 * k + 1] = w.y;
                }
            } else {
#pragma unroll
                for (int k = 0; k < 2 * D; ++k) ev[k] = (t0 + j2) * D + k < T * D ? ka.eps_t[(t0 + j2) * D + k] : 0.0;
            }
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int j = j2 + jj;
                double nx[D];
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    double v = ka.a[i];
#pragma unroll
                    for (int k = 0; k <= i; ++k) v = fma(ka.Lq[i][k], ev[jj * D + k], v);      // (lower factor)
                    nx[i] = v;
                }
                TGP_MATVEC_ACC(ka.A, x, nx);
                double yy = fma(ka.sR, ee[j], ka.hh);
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    x[i] = nx[i];
                    yy = fma(ka.h[i], nx[i], yy);
                }
                y0[j] = yy;
            }
        }
    }
    __syncthreads();      // (the table)
    // ---- inclusive scan of the lanes' end states: x_l <- sum_{m <= l} A^(8 (l - m)) x_m
    double st[D];
    if (any_valid) {
#define TGP_RAND_LEVEL(K)                                                                  \
    do {                                                                                   \
        double g_[D], n_[D];                                                               \
        _Pragma("unroll") for (int i = 0; i < D; ++i) {                                    \
            g_[i] = dpp_mov<0x110 + (1 << (K))>(x[i]);                                     \
            n_[i] = x[i];                                                                  \
        }                                                                                  \
        TGP_MATVEC_ACC(ka.P[K], g_, n_);                                                   \
        _Pragma("unroll") for (int i = 0; i < D; ++i) x[i] = n_[i];                        \
    } while (0)
        TGP_RAND_LEVEL(0);
        TGP_RAND_LEVEL(1);
        TGP_RAND_LEVEL(2);
        TGP_RAND_LEVEL(3);
#undef TGP_RAND_LEVEL
        {
            const int e1 = (lane & 15) + 1, e2 = lane >= 32 ? lane - 31 : 0;
            double gv[D], nv[D];
#pragma unroll
            for (int i = 0; i < D; ++i) {
                gv[i] = dpp_mov<0x142, 0xA>(x[i]);
                nv[i] = x[i];
            }
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int k = 0; k < D; ++k) nv[i] = fma(sPw[i][k][e1], gv[k], nv[i]);
#pragma unroll
            for (int i = 0; i < D; ++i) {
                x[i] = nv[i];
                gv[i] = dpp_mov<0x143, 0xC>(nv[i]);
            }
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int k = 0; k < D; ++k) nv[i] = fma(sPw[i][k][e2], gv[k], nv[i]);
#pragma unroll
            for (int i = 0; i < D; ++i) x[i] = nv[i];
        }
    }
#pragma unroll
    for (int i = 0; i < D; ++i) st[i] = dpp_mov<0x138>(x[i]);      // the state in front of the lane's steps (lane 0: zero)
    if (lane == 63) {
#pragma unroll
        for (int i = 0; i < D; ++i) sF[wave][i] = any_valid ? x[i] : 0.0;
    }
    __syncthreads();
    if (!any_valid) return;
    // ---- the tile's start state from the <= 3 tiles before it (workgroup 0: the drawn x0 sits at position -1), then the outputs
    double zin[D];
#pragma unroll
    for (int i = 0; i < D; ++i) zin[i] = 0.0;
#pragma unroll
    for (int k = 1; k <= 3; ++k) {
        const int src = wave - k;
        if (src < -1 || (src == -1 && !first)) continue;      // (wave-uniform)
        double xs[D];
#pragma unroll
        for (int i = 0; i < D; ++i) xs[i] = src >= 0 ? sF[src][i] : ka.x0[i];
        if (k == 1) {
#pragma unroll
            for (int i = 0; i < D; ++i) zin[i] += xs[i];
        } else {
            TGP_MATVEC_ACC(ka.PT[k - 2], xs, zin);
        }
    }
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int k = 0; k < D; ++k) st[i] = fma(sPw[i][k][lane], zin[k], st[i]);
    const bool aligned = (reinterpret_cast<uintptr_t>(ka.y) & 15) == 0;
#pragma unroll
    for (int j = 0; j < SUB; j += 2) {
        double m0 = y0[j], m1 = y0[j + 1];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            m0 = fma(ka.WJ[j][i], st[i], m0);
            m1 = fma(ka.WJ[j + 1][i], st[i], m1);
        }
        const long long t = t0 + j;
        if (t >= c_lo && t + 1 < c_hi && aligned) {
            v2d w;
            w.x = m0;
            w.y = m1;
            *reinterpret_cast<v2d*>(ka.y + t) = w;
        } else {
            if (t >= c_lo && t < c_hi) ka.y[t] = m0;
            if (t + 1 >= c_lo && t + 1 < c_hi) ka.y[t + 1] = m1;
        }
    }
}
#undef TGP_MATVEC_ACC

template <int D>
int launch_rand(hipStream_t st, const tgp_plan::RandPlan& rp, const double* x0, const double* eps_t, const double* eps_e, long long T, double* y) {
    constexpr int NW = 8, M = tgp_plan::kRandMaxD;
    RArgs<D> ka;
    static_assert(sizeof(RArgs<D>) <= 8192, "the kernel-argument segment (12 KB launch on gfx950: scripts/micro/bigarg.hip)");
    std::memset(&ka, 0, sizeof ka);
    for (int i = 0; i < D; ++i) {
        ka.a[i] = rp.a[i];
        ka.h[i] = rp.h[i];
        ka.x0[i] = x0[i];
        for (int k = 0; k < D; ++k) {
            ka.A[i][k] = rp.A[i * D + k];
            ka.Lq[i][k] = rp.Lq[i * D + k];
            for (int b = 0; b < 6; ++b) ka.P[b][i][k] = rp.P[b][i * D + k];
            for (int b = 0; b < 2; ++b) ka.PT[b][i][k] = rp.PT[b][i * D + k];
        }
    }
    (void)M;
    for (int j = 0; j < kWJ; ++j)
        for (int i = 0; i < D; ++i) ka.WJ[j][i] = rp.WJ[j][i];
    ka.hh = rp.hh;
    ka.sR = rp.sR;
    ka.T = T;
    ka.halo = rp.halo;
    ka.C = (long long)NW * 64 * kWJ - rp.halo;
    ka.nwg = (T + ka.C - 1) / ka.C;
    ka.eps_t = eps_t;
    ka.eps_e = eps_e;
    ka.y = y;
    const long long per = (ka.nwg + 7) / 8;
    hipLaunchKernelGGL((k_rand_one<D, NW>), dim3((unsigned)(per * 8)), dim3(NW * 64), 0, st, ka);
    return (int)hipGetLastError();
}
}  // namespace

int rand_lti(hipStream_t stream, const tgp_plan::RandPlan& plan, const double* x0, const double* eps_t, const double* eps_e, long long T, double* y,
             const char** kname) {
    if (kname) *kname = "k_rand_one";
    switch (plan.d) {
        case 1: return launch_rand<1>(stream, plan, x0, eps_t, eps_e, T, y);
        case 2: return launch_rand<2>(stream, plan, x0, eps_t, eps_e, T, y);
        case 3: return launch_rand<3>(stream, plan, x0, eps_t, eps_e, T, y);
        case 4: return launch_rand<4>(stream, plan, x0, eps_t, eps_e, T, y);
        case 5: return launch_rand<5>(stream, plan, x0, eps_t, eps_e, T, y);
        case 6: return launch_rand<6>(stream, plan, x0, eps_t, eps_e, T, y);
        case 7: return launch_rand<7>(stream, plan, x0, eps_t, eps_e, T, y);
        case 8: return launch_rand<8>(stream, plan, x0, eps_t, eps_e, T, y);
    }
    return (int)hipErrorInvalidValue;
}

// =================================================================================================================================
// _filter of an LTI model behind its head (lgssm.jl:171-187): mu' = Phi mu + a + (A K) u, r = u - h' mu, m_t = mu_t + K r_t, P_t = P_ss.
// k_rand_one's structure on the dense powers of the closed loop Phi = A - (A K) h': a first sweep of a lane's 8 steps from a zero state gives
// its end state, the scan and the tile chaining its true start state, a second sweep from there the outputs.  Workgroup 0 starts at nhs from
// the head's end state (computed on the host).  Reads y once, writes 8 (d + d^2) bytes per step.
// =================================================================================================================================
namespace {
template <int D>
struct FArgs {
    double Phi[D][D], a[D], kA[D], K[D], h[D], hh;
    double P[6][D][D], PT[2][D][D];
    double Pss[D * D];
    double Gss[D * D], Lss[D * D];      // posterior(model, y): the settled reverse-time transition and its noise, COLUMN-major as they leave
    double mu0[D];
    long long T, C, nwg, nhs;
    int halo;
    const double* y;
    double *m, *Pc, *part;
    double *Gc, *gc, *Lc, *fin;      // (all four or none) G, L [T][D D], g [T][D]; fin [D]: the last filtered mean
};

template <int D, int NW>
__global__ __launch_bounds__(NW * 64, (D <= 3 ? 4 : 2)) void k_filter_one(const FArgs<D> by_value) {      // (d = 4 at four waves per SIMD: 18 registers spilled)
    (void)by_value;
    const FArgs<D>& ka = *(const FArgs<D>*)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int SUB = kWJ, TILE = 64 * SUB;
    __shared__ double sF[NW][D], sAcc[NW];
    __shared__ double sPw[D][D][64];      // Phi^(8 e), e = 0 .. 63
    __shared__ double sP[3][D * D];       // the settled covariance, G, L (the fills below index them per lane)
    constexpr int LV = SUB * D / 2;        // 16-byte pieces of means per lane
    __shared__ v2d sT[NW][16 * (LV + 1)];      // a quarter of a tile's means (or g) on their way from the lanes to memory order
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long g;
    {
        const long long per = (ka.nwg + 7) / 8;
        g = (long long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
        if ((long long)(blockIdx.x >> 3) >= per || g >= ka.nwg) return;
    }
    if (threadIdx.x < D * D) {
        sP[0][threadIdx.x] = ka.Pss[threadIdx.x];
        sP[1][threadIdx.x] = ka.Gss[threadIdx.x];
        sP[2][threadIdx.x] = ka.Lss[threadIdx.x];
    }
    const long long T = ka.T, c_lo = ka.nhs + g * ka.C, c_hi = (c_lo + ka.C < T) ? c_lo + ka.C : T;
    const bool first = g == 0;
    const long long s0 = first ? ka.nhs : c_lo - ka.halo;
    const long long tile_t0 = s0 + (long long)wave * TILE, t0 = tile_t0 + (long long)lane * SUB;
    const bool any_valid = tile_t0 < c_hi;
    {
        const int uw = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
        for (int i = 0; i < D; ++i) {
            if (uw != i % NW) continue;
            double row[D];
#pragma unroll
            for (int k = 0; k < D; ++k) row[k] = (k == i) ? 1.0 : 0.0;
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                double nr[D];
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    double v = 0.0;
#pragma unroll
                    for (int m = 0; m < D; ++m) v = fma(row[m], ka.P[b][m][k], v);
                    nr[k] = v;
                }
                const bool bit = ((lane >> b) & 1) != 0;
#pragma unroll
                for (int k = 0; k < D; ++k) row[k] = bit ? nr[k] : row[k];
            }
#pragma unroll
            for (int k = 0; k < D; ++k) sPw[i][k][lane] = row[k];
        }
    }
    // ---- first sweep: the lane's 8 steps from a zero state
    double u[SUB], x[D];
#pragma unroll
    for (int i = 0; i < D; ++i) x[i] = 0.0;
#pragma unroll
    for (int j = 0; j < SUB; ++j) u[j] = 0.0;
    if (any_valid) {
        if (t0 + SUB <= T && (reinterpret_cast<uintptr_t>(ka.y) & 15) == 0) {
            const v2d* q = reinterpret_cast<const v2d*>(ka.y + t0);
#pragma unroll
            for (int j = 0; j < SUB / 2; ++j) {
                const v2d w = q[j];
                u[2 * j] = w.x - ka.hh;
                u[2 * j + 1] = w.y - ka.hh;
            }
        } else {
#pragma unroll
            for (int j = 0; j < SUB; ++j) u[j] = t0 + j < T ? ka.y[t0 + j] - ka.hh : 0.0;
        }
#pragma unroll
        for (int j = 0; j < SUB; ++j) {
            double nx[D];
#pragma unroll
            for (int i = 0; i < D; ++i) {
                double v = fma(ka.kA[i], u[j], ka.a[i]);
#pragma unroll
                for (int k = 0; k < D; ++k) v = fma(ka.Phi[i][k], x[k], v);
                nx[i] = v;
            }
#pragma unroll
            for (int i = 0; i < D; ++i) x[i] = nx[i];
        }
    }
    __syncthreads();
    double st[D];
    if (any_valid) {
#define TGP_FILT_LEVEL(K)                                                                  \
    do {                                                                                   \
        double g_[D], n_[D];                                                               \
        _Pragma("unroll") for (int i = 0; i < D; ++i) {                                    \
            g_[i] = dpp_mov<0x110 + (1 << (K))>(x[i]);                                     \
            n_[i] = x[i];                                                                  \
        }                                                                                  \
        _Pragma("unroll") for (int i = 0; i < D; ++i)                                      \
            _Pragma("unroll") for (int k = 0; k < D; ++k) n_[i] = fma(ka.P[K][i][k], g_[k], n_[i]); \
        _Pragma("unroll") for (int i = 0; i < D; ++i) x[i] = n_[i];                        \
    } while (0)
        TGP_FILT_LEVEL(0);
        TGP_FILT_LEVEL(1);
        TGP_FILT_LEVEL(2);
        TGP_FILT_LEVEL(3);
#undef TGP_FILT_LEVEL
        const int e1 = (lane & 15) + 1, e2 = lane >= 32 ? lane - 31 : 0;
        double gv[D], nv[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            gv[i] = dpp_mov<0x142, 0xA>(x[i]);
            nv[i] = x[i];
        }
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int k = 0; k < D; ++k) nv[i] = fma(sPw[i][k][e1], gv[k], nv[i]);
#pragma unroll
        for (int i = 0; i < D; ++i) {
            x[i] = nv[i];
            gv[i] = dpp_mov<0x143, 0xC>(nv[i]);
        }
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int k = 0; k < D; ++k) nv[i] = fma(sPw[i][k][e2], gv[k], nv[i]);
#pragma unroll
        for (int i = 0; i < D; ++i) x[i] = nv[i];
    }
#pragma unroll
    for (int i = 0; i < D; ++i) st[i] = dpp_mov<0x138>(x[i]);
    if (lane == 63) {
#pragma unroll
        for (int i = 0; i < D; ++i) sF[wave][i] = any_valid ? x[i] : 0.0;
    }
    __syncthreads();
    double acc = 0.0;
    if (any_valid) {
        double zin[D];
#pragma unroll
        for (int i = 0; i < D; ++i) zin[i] = 0.0;
#pragma unroll
        for (int k = 1; k <= 3; ++k) {
            const int src = wave - k;
            if (src < -1 || (src == -1 && !first)) continue;
            double xs[D];
#pragma unroll
            for (int i = 0; i < D; ++i) xs[i] = src >= 0 ? sF[src][i] : ka.mu0[i];
            if (k == 1) {
#pragma unroll
                for (int i = 0; i < D; ++i) zin[i] += xs[i];
            } else {
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int m = 0; m < D; ++m) zin[i] = fma(ka.PT[k - 2][i][m], xs[m], zin[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int k = 0; k < D; ++k) st[i] = fma(sPw[i][k][lane], zin[k], st[i]);
        // ---- second sweep from the true start state: innovations, filtered means
        const bool whole = t0 >= c_lo && t0 + SUB <= c_hi;
        const bool m_al = ka.m != nullptr && (reinterpret_cast<uintptr_t>(ka.m) & 15) == 0;
        // a tile wholly inside the workgroup's range (and short of the series' last step): its 512 D means -- or the 512 D values of g, one step
        // to the right -- are one run of consecutive doubles; they leave in memory order through LDS (below) instead of lane by lane
        const bool run_whole = tile_t0 >= c_lo && tile_t0 + TILE <= c_hi && tile_t0 + TILE < T;      // (wave-uniform)
        double* const run = !run_whole ? nullptr : (ka.gc != nullptr ? ka.gc + (tile_t0 + 1) * D : (ka.m != nullptr ? ka.m + tile_t0 * D : nullptr));
        const bool want_g = ka.gc != nullptr;
        double oall[SUB * D];
#pragma unroll
        for (int j2 = 0; j2 < SUB; j2 += 2) {
            double mf[2 * D];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int j = j2 + jj;
                double r = u[j];
#pragma unroll
                for (int k = 0; k < D; ++k) r = fma(-ka.h[k], st[k], r);
                const long long t = t0 + j;
                if (t >= c_lo && t < c_hi) acc = fma(r, r, acc);
                double nx[D];
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    mf[jj * D + i] = fma(ka.K[i], r, st[i]);
                    double v = fma(ka.kA[i], u[j], ka.a[i]);
#pragma unroll
                    for (int k = 0; k < D; ++k) v = fma(ka.Phi[i][k], st[k], v);
                    nx[i] = v;
                }
#pragma unroll
                for (int i = 0; i < D; ++i) st[i] = nx[i];
                if (run != nullptr) {
#pragma unroll
                    for (int i = 0; i < D; ++i) {
                        double v = mf[jj * D + i];
                        if (want_g) {
#pragma unroll
                            for (int k = 0; k < D; ++k) v = fma(-ka.Gss[k * D + i], nx[k], v);
                        }
                        oall[j * D + i] = v;
                    }
                } else if (ka.gc != nullptr && t >= c_lo && t < c_hi) {
                    // step t + 1 of the reverse-time model (lgssm.jl:215-221, :231-238): g = m_t - G mu_(t+1)
                    if (t + 1 < T) {
#pragma unroll
                        for (int i = 0; i < D; ++i) {
                            double v = mf[jj * D + i];
#pragma unroll
                            for (int k = 0; k < D; ++k) v = fma(-ka.Gss[k * D + i], nx[k], v);
                            ka.gc[(t + 1) * D + i] = v;
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < D; ++i) ka.fin[i] = mf[jj * D + i];
                    }
                }
            }
            if (ka.m != nullptr && run == nullptr) {
                const long long t = t0 + j2;
                if (whole && m_al) {      // (t D is even: 16-byte pieces)
                    v2d* q = reinterpret_cast<v2d*>(ka.m + t * D);
#pragma unroll
                    for (int k = 0; k < D; ++k) {
                        v2d w;
                        w.x = mf[2 * k];
                        w.y = mf[2 * k + 1];
                        q[k] = w;
                    }
                } else {
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
                        if (t + jj >= c_lo && t + jj < c_hi)
#pragma unroll
                            for (int i = 0; i < D; ++i) ka.m[(t + jj) * D + i] = mf[jj * D + i];
                }
            }
        }
        if (run != nullptr) {
            // sixteen lanes' values at a time: in at 16-byte pieces per lane (one spare piece per lane: no bank conflicts), out as doubles in
            // memory order -- every store instruction 512 consecutive bytes (g sits D doubles off the tile's start: 8-byte granularity)
            const double* sTd = reinterpret_cast<const double*>(&sT[wave][0]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if ((lane >> 4) == r) {
#pragma unroll
                    for (int k = 0; k < LV; ++k) {
                        v2d w;
                        w.x = oall[2 * k];
                        w.y = oall[2 * k + 1];
                        sT[wave][(lane & 15) * (LV + 1) + k] = w;
                    }
                }
                lds_sync();
#pragma unroll
                for (int i = 0; i < 2 * D; ++i) {
                    const int e = i * 64 + lane;      // of the 16 lanes' 128 D doubles
                    run[r * 128 * D + e] = sTd[e + 2 * (e / (SUB * D))];
                }
                lds_sync();
            }
        }
        // the covariances: the settled one, for every step the workgroup owns -- a whole tile as one run of 512 D^2 doubles, every store
        // instruction 1 KB of consecutive bytes (a lane writing its own 8 D^2 values, 16 bytes at a 64 D^2-byte stride: 40 % slower)
        constexpr int DD = D * D;
        const bool tile_whole = tile_t0 >= c_lo && tile_t0 + TILE <= c_hi;      // (wave-uniform)
#pragma unroll
        for (int which = 0; which < 3; ++which) {
            double* dst = which == 0 ? ka.Pc : (which == 1 ? ka.Gc : ka.Lc);
            if (dst == nullptr) continue;
            if (tile_whole && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
                v2d* q = reinterpret_cast<v2d*>(dst + tile_t0 * DD);
#pragma unroll 4
                for (int k = 0; k < SUB * DD / 2; ++k) {
                    const int e = k * 64 + lane;
                    v2d w;
                    w.x = sP[which][(2 * e) % DD];
                    w.y = sP[which][(2 * e + 1) % DD];
                    q[e] = w;
                }
            } else {
#pragma unroll
                for (int j = 0; j < SUB; ++j)
                    if (t0 + j >= c_lo && t0 + j < c_hi)
#pragma unroll
                        for (int e = 0; e < DD; ++e) dst[(t0 + j) * DD + e] = sP[which][e];
            }
        }
    }
    acc = wave_sum_to_lane63(acc);
    if (lane == 63) sAcc[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tsum = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) tsum += sAcc[w];
        ka.part[g] = tsum;
    }
}

template <int D>
int launch_filter(hipStream_t st, const tgp_plan::FilterPlan& fp, const double* mu0, const double* y, long long T, double* m, double* Pc, double* part,
                  const PosteriorOut* po) {
    constexpr int NW = 8;
    FArgs<D> ka;
    static_assert(sizeof(FArgs<D>) <= 8192, "the kernel-argument segment (12 KB launch on gfx950: scripts/micro/bigarg.hip)");
    std::memset(&ka, 0, sizeof ka);
    for (int i = 0; i < D; ++i) {
        ka.a[i] = fp.a[i];
        ka.kA[i] = fp.kA[i];
        ka.K[i] = fp.K[i];
        ka.h[i] = fp.h[i];
        ka.mu0[i] = mu0[i];
        for (int k = 0; k < D; ++k) {
            ka.Phi[i][k] = fp.Phi[i * D + k];
            ka.Pss[i * D + k] = fp.Pss[i * D + k];
            if (po) {
                ka.Gss[i * D + k] = po->Gss[i * D + k];
                ka.Lss[i * D + k] = po->Lss[i * D + k];
            }
            for (int b = 0; b < 6; ++b) ka.P[b][i][k] = fp.P[b][i * D + k];
            for (int b = 0; b < 2; ++b) ka.PT[b][i][k] = fp.PT[b][i * D + k];
        }
    }
    ka.hh = fp.hh;
    ka.T = T;
    ka.nhs = fp.nhs;
    ka.halo = fp.halo;
    ka.C = (long long)NW * 64 * kWJ - fp.halo;
    ka.nwg = (T - fp.nhs + ka.C - 1) / ka.C;
    ka.y = y;
    ka.m = m;
    ka.Pc = Pc;
    ka.part = part;
    if (po) {
        ka.Gc = po->G;
        ka.gc = po->g;
        ka.Lc = po->L;
        ka.fin = po->fin;
    }
    const long long per = (ka.nwg + 7) / 8;
    hipLaunchKernelGGL((k_filter_one<D, NW>), dim3((unsigned)(per * 8)), dim3(NW * 64), 0, st, ka);
    return (int)hipGetLastError();
}
}  // namespace

long long filter_workgroups(const tgp_plan::FilterPlan& fp, long long T) {
    const long long C = 8LL * 64 * kWJ - fp.halo;
    return (T - fp.nhs + C - 1) / C;
}

int filter_lti(hipStream_t stream, const tgp_plan::FilterPlan& fp, const double* mu_start, const double* y, long long T, double* m_out, double* P_out,
               double* part, const PosteriorOut* po) {
    switch (fp.d) {
        case 1: return launch_filter<1>(stream, fp, mu_start, y, T, m_out, P_out, part, po);
        case 2: return launch_filter<2>(stream, fp, mu_start, y, T, m_out, P_out, part, po);
        case 3: return launch_filter<3>(stream, fp, mu_start, y, T, m_out, P_out, part, po);
        case 4: return launch_filter<4>(stream, fp, mu_start, y, T, m_out, P_out, part, po);
        case 5: return launch_filter<5>(stream, fp, mu_start, y, T, m_out, P_out, part, po);
        case 6: return launch_filter<6>(stream, fp, mu_start, y, T, m_out, P_out, part, po);
        case 7: return launch_filter<7>(stream, fp, mu_start, y, T, m_out, P_out, part, po);
        case 8: return launch_filter<8>(stream, fp, mu_start, y, T, m_out, P_out, part, po);
    }
    return (int)hipErrorInvalidValue;
}

// =================================================================================================================================
// The adjoint (gradient) pass of an LTI model behind its head, in one launch.  k_filter_one's forward half gives a lane
// its true start state; a second forward sweep keeps mu_j and r_j of its 8 steps; the adjoint psi = Phi' psi + h r / S runs the same way
// backwards (zero-start sweep, reverse scan by DPP moves on the TRANSPOSED powers -- the same table read the other way --, chaining over the
// <= 3 tiles behind); the last sweep walks the 8 steps backwards with the true psi and accumulates the d^2 + 3 d + 2 sums.  Spans carry a
// halo at both ends.
// =================================================================================================================================
namespace {
template <int D>
struct GArgs {
    double Phi[D][D], a[D], kA[D], h[D], hh, iS;
    double P[6][D][D], PT[2][D][D];
    double mu0[D];
    long long T, C, nwg, nhs;
    int halo;
    const double* y;
    double *part, *psi_out;
    // the head beside the kernel (SmoothCall in tgp_modal.hpp): workgroup 0 hands the head's observations over through pinned memory and waits
    // (bounded) for the predicted mean of step nhs; nullptr: mu0 above by value
    double* head_in;
    long long* head_in_flag;
    const double* mu0p;
    const long long* mu0_flag;
    long long seq;
};

template <int D, int NW>
__global__ __launch_bounds__(NW * 64, 2) void k_adjoint_one(const GArgs<D> by_value) {
    (void)by_value;
    const GArgs<D>& ka = *(const GArgs<D>*)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int SUB = kWJ, TILE = 64 * SUB, NS = D * D + 3 * D + 2;
    __shared__ double sF[NW][D], sB[NW][D], sAcc[NW][NS];
    __shared__ double sPw[D][D][64];      // Phi^(8 e), e = 0 .. 63
    __shared__ double sPoison;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) sPoison = 0.0;
    long long g;
    {
        const long long per = (ka.nwg + 7) / 8;
        g = (long long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
        if ((long long)(blockIdx.x >> 3) >= per || g >= ka.nwg) return;
    }
    const long long T = ka.T, c_lo = ka.nhs + g * ka.C, c_hi = (c_lo + ka.C < T) ? c_lo + ka.C : T;
    const bool first = g == 0;
    const bool from_head = first || c_lo - ka.halo < ka.nhs;      // (a span shorter than a halo: the second workgroup starts behind the head too -- see k_smooth_one)
    const long long s0 = from_head ? ka.nhs : c_lo - ka.halo;
    const long long tile_t0 = s0 + (long long)wave * TILE, t0 = tile_t0 + (long long)lane * SUB;
    const bool any_valid = tile_t0 < T && tile_t0 < c_hi + ka.halo;      // (wave-uniform: tiles behind the right-hand halo have nothing to do)
    if (first && wave == NW - 1 && ka.head_in != nullptr) {      // the head's observations to the host, first thing
        for (long long t = lane; t < ka.nhs; t += 64) ka.head_in[t] = ka.y[t];
        __threadfence_system();
        if (lane == 0) __hip_atomic_store(ka.head_in_flag, 2 * ka.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    {
        const int uw = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
        for (int i = 0; i < D; ++i) {
            if (uw != i % NW) continue;
            double row[D];
#pragma unroll
            for (int k = 0; k < D; ++k) row[k] = (k == i) ? 1.0 : 0.0;
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                double nr[D];
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    double v = 0.0;
#pragma unroll
                    for (int m = 0; m < D; ++m) v = fma(row[m], ka.P[b][m][k], v);
                    nr[k] = v;
                }
                const bool bit = ((lane >> b) & 1) != 0;
#pragma unroll
                for (int k = 0; k < D; ++k) row[k] = bit ? nr[k] : row[k];
            }
#pragma unroll
            for (int k = 0; k < D; ++k) sPw[i][k][lane] = row[k];
        }
    }
    // ---- forward, zero start
    double u[SUB], x[D];
#pragma unroll
    for (int i = 0; i < D; ++i) x[i] = 0.0;
#pragma unroll
    for (int j = 0; j < SUB; ++j) u[j] = 0.0;
    if (any_valid) {
        if (t0 + SUB <= T && (reinterpret_cast<uintptr_t>(ka.y) & 15) == 0) {
            const v2d* q = reinterpret_cast<const v2d*>(ka.y + t0);
#pragma unroll
            for (int j = 0; j < SUB / 2; ++j) {
                const v2d w = q[j];
                u[2 * j] = w.x - ka.hh;
                u[2 * j + 1] = w.y - ka.hh;
            }
        } else {
#pragma unroll
            for (int j = 0; j < SUB; ++j) u[j] = t0 + j < T ? ka.y[t0 + j] - ka.hh : 0.0;
        }
#pragma unroll
        for (int j = 0; j < SUB; ++j) {
            double nx[D];
#pragma unroll
            for (int i = 0; i < D; ++i) {
                double v = fma(ka.kA[i], u[j], ka.a[i]);
#pragma unroll
                for (int k = 0; k < D; ++k) v = fma(ka.Phi[i][k], x[k], v);
                nx[i] = v;
            }
#pragma unroll
            for (int i = 0; i < D; ++i) x[i] = nx[i];
        }
    }
    __syncthreads();
    double st[D];
    if (any_valid) {
#define TGP_ADJ_FWD(K)                                                                     \
    do {                                                                                   \
        double g_[D], n_[D];                                                               \
        _Pragma("unroll") for (int i = 0; i < D; ++i) {                                    \
            g_[i] = dpp_mov<0x110 + (1 << (K))>(x[i]);                                     \
            n_[i] = x[i];                                                                  \
        }                                                                                  \
        _Pragma("unroll") for (int i = 0; i < D; ++i)                                      \
            _Pragma("unroll") for (int k = 0; k < D; ++k) n_[i] = fma(ka.P[K][i][k], g_[k], n_[i]); \
        _Pragma("unroll") for (int i = 0; i < D; ++i) x[i] = n_[i];                        \
    } while (0)
        TGP_ADJ_FWD(0);
        TGP_ADJ_FWD(1);
        TGP_ADJ_FWD(2);
        TGP_ADJ_FWD(3);
#undef TGP_ADJ_FWD
        const int e1 = (lane & 15) + 1, e2 = lane >= 32 ? lane - 31 : 0;
        double gv[D], nv[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            gv[i] = dpp_mov<0x142, 0xA>(x[i]);
            nv[i] = x[i];
        }
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int k = 0; k < D; ++k) nv[i] = fma(sPw[i][k][e1], gv[k], nv[i]);
#pragma unroll
        for (int i = 0; i < D; ++i) {
            x[i] = nv[i];
            gv[i] = dpp_mov<0x143, 0xC>(nv[i]);
        }
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int k = 0; k < D; ++k) nv[i] = fma(sPw[i][k][e2], gv[k], nv[i]);
#pragma unroll
        for (int i = 0; i < D; ++i) x[i] = nv[i];
    }
#pragma unroll
    for (int i = 0; i < D; ++i) st[i] = dpp_mov<0x138>(x[i]);
    if (lane == 63) {
#pragma unroll
        for (int i = 0; i < D; ++i) sF[wave][i] = any_valid ? x[i] : 0.0;
    }
    __syncthreads();
    // ---- the true start state, then the lane's predicted means and innovations
    double mus[SUB][D], r[SUB], psi[D];
#pragma unroll
    for (int i = 0; i < D; ++i) psi[i] = 0.0;
#pragma unroll
    for (int j = 0; j < SUB; ++j) {
        r[j] = 0.0;
#pragma unroll
        for (int i = 0; i < D; ++i) mus[j][i] = 0.0;
    }
    if (any_valid) {
        double zin[D];
#pragma unroll
        for (int i = 0; i < D; ++i) zin[i] = 0.0;
        const bool from_host = from_head && wave < 3 && ka.mu0_flag != nullptr;      // (wave-uniform) the head's end state comes from the host, beside the kernel
        if (from_host) wait_tables(ka.mu0_flag, ka.seq, &sPoison);
#pragma unroll
        for (int k = 1; k <= 3; ++k) {
            const int src = wave - k;
            if (src < -1 || (src == -1 && !from_head)) continue;
            double xs[D];
#pragma unroll
            for (int i = 0; i < D; ++i) xs[i] = src >= 0 ? sF[src][i] : (from_host ? ka.mu0p[i] : ka.mu0[i]);
            if (k == 1) {
#pragma unroll
                for (int i = 0; i < D; ++i) zin[i] += xs[i];
            } else {
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int m = 0; m < D; ++m) zin[i] = fma(ka.PT[k - 2][i][m], xs[m], zin[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int k = 0; k < D; ++k) st[i] = fma(sPw[i][k][lane], zin[k], st[i]);
#pragma unroll
        for (int j = 0; j < SUB; ++j) {
            double rr = u[j];
#pragma unroll
            for (int k = 0; k < D; ++k) {
                mus[j][k] = st[k];
                rr = fma(-ka.h[k], st[k], rr);
            }
            r[j] = t0 + j < T ? rr : 0.0;      // (steps behind the series' end: no innovation, no adjoint)
            double nx[D];
#pragma unroll
            for (int i = 0; i < D; ++i) {
                double v = fma(ka.kA[i], u[j], ka.a[i]);
#pragma unroll
                for (int k = 0; k < D; ++k) v = fma(ka.Phi[i][k], st[k], v);
                nx[i] = v;
            }
#pragma unroll
            for (int i = 0; i < D; ++i) st[i] = nx[i];
        }
        // ---- backward, zero psi behind the lane: psi <- Phi' psi + h r / S
#pragma unroll
        for (int j = SUB - 1; j >= 0; --j) {
            const double c = r[j] * ka.iS;
            double np[D];
#pragma unroll
            for (int i = 0; i < D; ++i) {
                double v = ka.h[i] * c;
#pragma unroll
                for (int k = 0; k < D; ++k) v = fma(ka.Phi[k][i], psi[k], v);
                np[i] = v;
            }
#pragma unroll
            for (int i = 0; i < D; ++i) psi[i] = np[i];
        }
        // reverse scan: lane l <- sum_{m >= l} (Phi')^(8 (m - l)) psi_m
#define TGP_ADJ_BWD(K)                                                                     \
    do {                                                                                   \
        double g_[D], n_[D];                                                               \
        _Pragma("unroll") for (int i = 0; i < D; ++i) {                                    \
            g_[i] = dpp_mov<0x100 + (1 << (K))>(psi[i]);                                   \
            n_[i] = psi[i];                                                                \
        }                                                                                  \
        _Pragma("unroll") for (int i = 0; i < D; ++i)                                      \
            _Pragma("unroll") for (int k = 0; k < D; ++k) n_[i] = fma(ka.P[K][k][i], g_[k], n_[i]); \
        _Pragma("unroll") for (int i = 0; i < D; ++i) psi[i] = n_[i];                      \
    } while (0)
        TGP_ADJ_BWD(0);
        TGP_ADJ_BWD(1);
        TGP_ADJ_BWD(2);
        TGP_ADJ_BWD(3);
#undef TGP_ADJ_BWD
        {
            const int e1 = 16 - (lane & 15);      // rows 0 and 2 take the first lane of the row above them, 16 - p lanes away
            double gv[D], nv[D];
#pragma unroll
            for (int i = 0; i < D; ++i) {
                gv[i] = dpp_mov<0x15F, 0x5>(dpp_mov<0x130>(psi[i]));
                nv[i] = psi[i];
            }
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int k = 0; k < D; ++k) nv[i] = fma(sPw[k][i][e1], gv[k], nv[i]);
#pragma unroll
            for (int i = 0; i < D; ++i) psi[i] = nv[i];
            const int e2 = lane < 32 ? 32 - lane : 0;      // the lower half takes lane 32
            const double keep = lane < 32 ? 1.0 : 0.0;
#pragma unroll
            for (int i = 0; i < D; ++i) gv[i] = readlane_d(psi[i], 32) * keep;
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int k = 0; k < D; ++k) nv[i] = fma(sPw[k][i][e2], gv[k], nv[i]);
#pragma unroll
            for (int i = 0; i < D; ++i) psi[i] = nv[i];
        }
    }
    double pin[D];      // the adjoint behind the lane's last step: its right neighbour's (lane 63: zero)
#pragma unroll
    for (int i = 0; i < D; ++i) pin[i] = dpp_mov<0x130>(psi[i]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < D; ++i) sB[wave][i] = any_valid ? psi[i] : 0.0;
    }
    __syncthreads();
    double acc[NS];
#pragma unroll
    for (int e = 0; e < NS; ++e) acc[e] = 0.0;
    if (any_valid) {
        double zin[D];
#pragma unroll
        for (int i = 0; i < D; ++i) zin[i] = 0.0;
#pragma unroll
        for (int k = 1; k <= 3; ++k) {
            const int src = wave + k;
            if (src >= NW) continue;
            if (k == 1) {
#pragma unroll
                for (int i = 0; i < D; ++i) zin[i] += sB[src][i];
            } else {
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int m = 0; m < D; ++m) zin[i] = fma(ka.PT[k - 2][m][i], sB[src][m], zin[i]);
            }
        }
        const int eb = 63 - lane;
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int k = 0; k < D; ++k) pin[i] = fma(sPw[k][i][eb], zin[k], pin[i]);
        // ---- the last sweep: the lane's steps backwards with the true adjoint; sums over the steps the workgroup owns
#pragma unroll
        for (int j = SUB - 1; j >= 0; --j) {
            const long long t = t0 + j;
            const double own = (t >= c_lo && t < c_hi) ? 1.0 : 0.0;
            const double rj = r[j];
#pragma unroll
            for (int i = 0; i < D; ++i) {
                const double pw = pin[i] * own;
#pragma unroll
                for (int k = 0; k < D; ++k) acc[i * D + k] = fma(pw, mus[j][k], acc[i * D + k]);
                acc[D * D + i] += pw;
                acc[D * D + D + i] = fma(pw, rj, acc[D * D + D + i]);
                acc[D * D + 2 * D + i] = fma(rj * own, mus[j][i], acc[D * D + 2 * D + i]);
            }
            acc[D * D + 3 * D] = fma(rj, own, acc[D * D + 3 * D]);
            acc[D * D + 3 * D + 1] = fma(rj * own, rj, acc[D * D + 3 * D + 1]);
            const double c = rj * ka.iS;
            double np[D];
#pragma unroll
            for (int i = 0; i < D; ++i) {
                double v = ka.h[i] * c;
#pragma unroll
                for (int k = 0; k < D; ++k) v = fma(ka.Phi[k][i], pin[k], v);
                np[i] = v;
            }
#pragma unroll
            for (int i = 0; i < D; ++i) pin[i] = np[i];
        }
        if (first && wave == 0 && lane == 0) {
#pragma unroll
            for (int i = 0; i < D; ++i) ka.psi_out[i] = pin[i];      // d logpdf / d mu at step nhs: where the host's half takes over
        }
    }
#pragma unroll
    for (int e = 0; e < NS; ++e) {
        const double s = wave_sum_to_lane63(acc[e]);
        if (lane == 63) sAcc[wave][e] = s;
    }
    __syncthreads();
    if (threadIdx.x < NS) {
        double tsum = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) tsum += sAcc[w][threadIdx.x];
        ka.part[g * NS + threadIdx.x] = tsum + sPoison;      // (NaN once a bounded wait for the host ran out: the call's result is discarded)
    }
}

template <int D>
int launch_adjoint(hipStream_t st, const tgp_plan::FilterPlan& fp, const double* mu0, const double* y, long long T, double* part, double* psi_out, const HeadHandover* hh) {
    constexpr int NW = 8;
    GArgs<D> ka;
    static_assert(sizeof(GArgs<D>) <= 4096, "the kernel-argument segment");
    std::memset(&ka, 0, sizeof ka);
    for (int i = 0; i < D; ++i) {
        ka.a[i] = fp.a[i];
        ka.kA[i] = fp.kA[i];
        ka.h[i] = fp.h[i];
        ka.mu0[i] = mu0 != nullptr ? mu0[i] : 0.0;
        for (int k = 0; k < D; ++k) {
            ka.Phi[i][k] = fp.Phi[i * D + k];
            for (int b = 0; b < 6; ++b) ka.P[b][i][k] = fp.P[b][i * D + k];
            for (int b = 0; b < 2; ++b) ka.PT[b][i][k] = fp.PT[b][i * D + k];
        }
    }
    ka.hh = fp.hh;
    ka.iS = fp.iS;
    ka.T = T;
    ka.nhs = fp.nhs;
    ka.halo = fp.halo;
    ka.C = (long long)NW * 64 * kWJ - 2LL * fp.halo;
    ka.nwg = (T - fp.nhs + ka.C - 1) / ka.C;
    ka.y = y;
    ka.part = part;
    ka.psi_out = psi_out;
    if (hh != nullptr) {
        ka.head_in = hh->head_in;
        ka.head_in_flag = hh->head_in_flag;
        ka.mu0p = hh->mu0;
        ka.mu0_flag = hh->mu0_flag;
        ka.seq = hh->seq;
    }
    const long long per = (ka.nwg + 7) / 8;
    hipLaunchKernelGGL((k_adjoint_one<D, NW>), dim3((unsigned)(per * 8)), dim3(NW * 64), 0, st, ka);
    return (int)hipGetLastError();
}
}  // namespace

long long adjoint_workgroups(const tgp_plan::FilterPlan& fp, long long T) {
    const long long C = 8LL * 64 * kWJ - 2LL * fp.halo;
    return C > 0 ? (T - fp.nhs + C - 1) / C : -1;
}

int adjoint_lti(hipStream_t stream, const tgp_plan::FilterPlan& fp, const double* mu_start, const double* y, long long T, double* part, double* psi_out,
                const HeadHandover* hh) {
    switch (fp.d) {
        case 1: return launch_adjoint<1>(stream, fp, mu_start, y, T, part, psi_out, hh);
        case 2: return launch_adjoint<2>(stream, fp, mu_start, y, T, part, psi_out, hh);
        case 3: return launch_adjoint<3>(stream, fp, mu_start, y, T, part, psi_out, hh);
        case 4: return launch_adjoint<4>(stream, fp, mu_start, y, T, part, psi_out, hh);
        case 5: return launch_adjoint<5>(stream, fp, mu_start, y, T, part, psi_out, hh);
        case 6: return launch_adjoint<6>(stream, fp, mu_start, y, T, part, psi_out, hh);
    }
    return (int)hipErrorInvalidValue;
}

// =================================================================================================================================
// logpdf + posterior marginals of an LTI model behind its head on DENSE powers in BOTH directions (round 5): the models the modal
// plan declines (a defective closed loop: two summands with one length scale, ...).  k_adjoint_one's skeleton -- spans with a halo at both ends,
// forward sweep from zero + DPP scan on the powers of Phi, reverse sweep from zero + reverse DPP scan, tiles chained through LDS -- with the
// smoother's recursion in the innovations going backwards, xi_t = c r_t + G xi_(t+1), on the powers of G (a second per-lane table), and no
// second sweep in either direction: the lane's true start state reaches its innovations through the rows h' Phi^j, its right-hand xi reaches
// its outputs through the rows h' G^(7-j) (the WJ / WG trick of k_steady_one, dense).  mean_t = y_t - (R/S) r_t + h' xi_(t+1);
// var_t = h' Ps h + R_new (constant but for the last n1 steps: tvb, pinned host memory, read in place by the last tiles).
// =================================================================================================================================
namespace {
template <int D>
struct SArgs {
    double Phi[D][D], a[D], kA[D], h[D], hh, rS, vb;
    double P[6][D][D], PT[2][D][D];
    double G[D][D], c[D];
    double GP[6][D][D], GPT[2][D][D];
    double WJ[kWJ][D], WG[kWJ][D];
    double mu0[D];
    long long T, C, nwg, nhs, n1;
    int halo, halo_r, rnew_per_step;      // halo_r: the halo behind a span (0 without the backward half)
    const double *y, *Rnew, *tvb;
    double *mean, *var, *part, *xi_out;
    // the head beside the kernel (tgp_modal.hpp SmoothCall): all pinned host memory, flags hold 2 seq once raised
    const double* mu0p;
    const long long* mu0_flag;
    long long* xi_flag;
    double* head_in;
    long long* head_in_flag;
    const double* head_out;
    const long long* head_out_flag;
    long long seq;
    // RAND: a draw from the posterior instead of its marginals (lgssm.jl:65-91 on the reverse-time model of :193-221).  With delta_t = x_t - m_t the
    // reverse-time step x_(t-1) = G x_t + g_t + U' eps_t is the smoother's recursion with a noise input: xi_t = c r_t + G xi_(t+1) + U' eps_t,
    // y_t = y_obs_t - (R / S) r_t + h' xi_(t+1) + sqrt(Rnew_t) eta_t.  U: the upper Cholesky factor of the settled L + 1e-9 I (U[k][i], k <= i);
    // xi_T = U0' eps_0 (the draw of the final filtering state) enters step T - 1 as v0 = G xi_T and its output as s0 = h' xi_T.
    const double* hhT;      // the emission offset per step (a mean function at the inputs, lti_sde.jl:118-131), or nullptr: hh above.  The gains do not see it.
    const double *eps_t, *eps_e;
    static constexpr int RD = D <= kSmoothRandMaxD ? D : 1;      // (the kernel-argument segment is full at d = 8: no room for what d > 4 never uses)
    double U[RD][RD], v0[RD], s0;
};

template <int D, int NW, int MINW, bool RAND>
__global__ __launch_bounds__(NW * 64, MINW) void k_smooth_one(const SArgs<D> by_value) {      // (RAND: `mean` receives the draw, `var` is unused)
    (void)by_value;
    const SArgs<D>& ka = *(const SArgs<D>*)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int SUB = kWJ, TILE = 64 * SUB;
    __shared__ double sF[NW][D], sB[NW][D], sAcc[NW];
    __shared__ double sPw[D][D][64];       // Phi^(8 e), e = 0 .. 63
    __shared__ double sPg[D][D][64];       // G^(8 e)
    __shared__ double sPoison;             // NaN once a bounded wait ran out (wait_tables): the call's result is then discarded
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) sPoison = 0.0;
    long long g;
    {
        const long long per = (ka.nwg + 7) / 8;
        g = (long long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
        if ((long long)(blockIdx.x >> 3) >= per || g >= ka.nwg) return;
    }
    const long long T = ka.T, c_lo = ka.nhs + g * ka.C, c_hi = (c_lo + ka.C < T) ? c_lo + ka.C : T;
    const bool first = g == 0;
    // a workgroup whose run-in would reach back into the head (a span shorter than a halo: the second workgroup) starts behind the head like the
    // first, from the head's exact end state: its tiles still cover its range and the halo behind it
    const bool from_head = first || c_lo - ka.halo < ka.nhs;
    const long long s0 = from_head ? ka.nhs : c_lo - ka.halo;
    const long long tile_t0 = s0 + (long long)wave * TILE, t0 = tile_t0 + (long long)lane * SUB;
    const bool any_valid = tile_t0 < T && tile_t0 < c_hi + ka.halo_r;      // (wave-uniform: tiles behind the right-hand halo have nothing to do)
    if (first && wave == NW - 1 && ka.head_in != nullptr) {      // the head's inputs to the host, first thing: its forward recursion runs there
        for (long long t = lane; t < ka.nhs; t += 64) {
            ka.head_in[t] = ka.y[t];
            if (ka.rnew_per_step && ka.mean != nullptr) ka.head_in[ka.nhs + t] = ka.Rnew[t];
        }
        if (!ka.rnew_per_step && ka.mean != nullptr && lane == 0) ka.head_in[ka.nhs] = ka.Rnew[0];
        if constexpr (RAND) {      // the head's draws: eta [nhs] behind y | Rnew, then eps [nhs][D]
            for (long long t = lane; t < ka.nhs; t += 64) ka.head_in[2 * ka.nhs + t] = ka.eps_e[t];
            for (long long e = lane; e < ka.nhs * D; e += 64) ka.head_in[3 * ka.nhs + e] = ka.eps_t[e];
        }
        if (ka.hhT != nullptr) {      // the head's emission offsets, behind everything else (offset 10 nhs)
            for (long long t = lane; t < ka.nhs; t += 64) ka.head_in[10 * ka.nhs + t] = ka.hhT[t];
        }
        __threadfence_system();
        if (lane == 0) __hip_atomic_store(ka.head_in_flag, 2 * ka.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // ---- the observations first (they are on their way while the tables are built)
    double u[SUB];
#pragma unroll
    for (int j = 0; j < SUB; ++j) u[j] = 0.0;
    if (any_valid) {
        if (ka.hhT != nullptr) {      // (kernel-uniform) an emission offset per step
#pragma unroll
            for (int j = 0; j < SUB; ++j) u[j] = t0 + j < T ? ka.y[t0 + j] - ka.hhT[t0 + j] : 0.0;
        } else if (t0 + SUB <= T && (reinterpret_cast<uintptr_t>(ka.y) & 15) == 0) {
            const v2d* q = reinterpret_cast<const v2d*>(ka.y + t0);
#pragma unroll
            for (int j = 0; j < SUB / 2; ++j) {
                const v2d w = q[j];
                u[2 * j] = w.x - ka.hh;
                u[2 * j + 1] = w.y - ka.hh;
            }
        } else {
#pragma unroll
            for (int j = 0; j < SUB; ++j) u[j] = t0 + j < T ? ka.y[t0 + j] - ka.hh : 0.0;
        }
    }
    {   // the per-lane power tables: row i of Phi^(8 lane) and of G^(8 lane), the 2 D rows dealt over the waves
        const int uw = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
        for (int i2 = 0; i2 < 2 * D; ++i2) {
            if (uw != i2 % NW) continue;
            const int i = i2 % D;
            const bool isg = i2 >= D;
            if (isg && ka.mean == nullptr) continue;      // (logpdf only: no backward half)
            double row[D];
#pragma unroll
            for (int k = 0; k < D; ++k) row[k] = (k == i) ? 1.0 : 0.0;
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                double nr[D];
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    double v = 0.0;
#pragma unroll
                    for (int m = 0; m < D; ++m) v = fma(row[m], isg ? ka.GP[b][m][k] : ka.P[b][m][k], v);
                    nr[k] = v;
                }
                const bool bit = ((lane >> b) & 1) != 0;
#pragma unroll
                for (int k = 0; k < D; ++k) row[k] = bit ? nr[k] : row[k];
            }
            if (isg) {
#pragma unroll
                for (int k = 0; k < D; ++k) sPg[i][k][lane] = row[k];
            } else {
#pragma unroll
                for (int k = 0; k < D; ++k) sPw[i][k][lane] = row[k];
            }
        }
    }
    // ---- forward, zero start: the lane's innovations r0 and its end state
    double r[SUB], x[D];
#pragma unroll
    for (int i = 0; i < D; ++i) x[i] = 0.0;
#pragma unroll
    for (int j = 0; j < SUB; ++j) r[j] = 0.0;
    if (any_valid) {
#pragma unroll
        for (int j = 0; j < SUB; ++j) {
            double rr = u[j];
#pragma unroll
            for (int k = 0; k < D; ++k) rr = fma(-ka.h[k], x[k], rr);
            r[j] = rr;
            double nx[D];
#pragma unroll
            for (int i = 0; i < D; ++i) {
                double v = fma(ka.kA[i], u[j], ka.a[i]);
#pragma unroll
                for (int k = 0; k < D; ++k) v = fma(ka.Phi[i][k], x[k], v);
                nx[i] = v;
            }
#pragma unroll
            for (int i = 0; i < D; ++i) x[i] = nx[i];
        }
    }
    __syncthreads();
    double st[D];
    if (any_valid) {
#define TGP_SM_FWD(K)                                                                      \
    do {                                                                                   \
        double g_[D], n_[D];                                                               \
        _Pragma("unroll") for (int i = 0; i < D; ++i) {                                    \
            g_[i] = dpp_mov<0x110 + (1 << (K))>(x[i]);                                     \
            n_[i] = x[i];                                                                  \
        }                                                                                  \
        _Pragma("unroll") for (int i = 0; i < D; ++i)                                      \
            _Pragma("unroll") for (int k = 0; k < D; ++k) n_[i] = fma(ka.P[K][i][k], g_[k], n_[i]); \
        _Pragma("unroll") for (int i = 0; i < D; ++i) x[i] = n_[i];                        \
    } while (0)
        TGP_SM_FWD(0);
        TGP_SM_FWD(1);
        TGP_SM_FWD(2);
        TGP_SM_FWD(3);
#undef TGP_SM_FWD
        const int e1 = (lane & 15) + 1, e2 = lane >= 32 ? lane - 31 : 0;
        double gv[D], nv[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            gv[i] = dpp_mov<0x142, 0xA>(x[i]);
            nv[i] = x[i];
        }
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int k = 0; k < D; ++k) nv[i] = fma(sPw[i][k][e1], gv[k], nv[i]);
#pragma unroll
        for (int i = 0; i < D; ++i) {
            x[i] = nv[i];
            gv[i] = dpp_mov<0x143, 0xC>(nv[i]);
        }
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int k = 0; k < D; ++k) nv[i] = fma(sPw[i][k][e2], gv[k], nv[i]);
#pragma unroll
        for (int i = 0; i < D; ++i) x[i] = nv[i];
    }
#pragma unroll
    for (int i = 0; i < D; ++i) st[i] = dpp_mov<0x138>(x[i]);
    if (lane == 63) {
#pragma unroll
        for (int i = 0; i < D; ++i) sF[wave][i] = any_valid ? x[i] : 0.0;
    }
    __syncthreads();
    // ---- the true start state -> the true innovations (r = r0 - h' Phi^j st); the reverse sweep from zero
    double xi[D], o0[SUB], acc = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i) xi[i] = 0.0;
#pragma unroll
    for (int j = 0; j < SUB; ++j) o0[j] = 0.0;
    if (any_valid) {
        double zin[D], mu0v[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            zin[i] = 0.0;
            mu0v[i] = ka.mu0[i];
        }
        if (from_head && wave < 3 && ka.mu0_flag != nullptr) {      // (wave-uniform) the head's end state comes from the host, beside the kernel
            wait_tables(ka.mu0_flag, ka.seq, &sPoison);
#pragma unroll
            for (int i = 0; i < D; ++i) mu0v[i] = ka.mu0p[i];
        }
#pragma unroll
        for (int k = 1; k <= 3; ++k) {
            const int src = wave - k;
            if (src < -1 || (src == -1 && !from_head)) continue;
            double xs[D];
#pragma unroll
            for (int i = 0; i < D; ++i) xs[i] = src >= 0 ? sF[src][i] : mu0v[i];
            if (k == 1) {
#pragma unroll
                for (int i = 0; i < D; ++i) zin[i] += xs[i];
            } else {
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int m = 0; m < D; ++m) zin[i] = fma(ka.PT[k - 2][i][m], xs[m], zin[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int k = 0; k < D; ++k) st[i] = fma(sPw[i][k][lane], zin[k], st[i]);
#pragma unroll
        for (int j = 0; j < SUB; ++j) {
            double rr = r[j];
#pragma unroll
            for (int k = 0; k < D; ++k) rr = fma(-ka.WJ[j][k], st[k], rr);
            const long long t = t0 + j;
            rr = t < T ? rr : 0.0;      // (steps behind the series' end: no innovation)
            r[j] = rr;
            if (t >= c_lo && t < c_hi) acc = fma(rr, rr, acc);
            // h' m_t + hh_t = y_t - (R / S) r_t: what the smoother adds h' xi_(t+1) to  (a per-step offset is fetched again: the L2 has it)
            o0[j] = fma(-ka.rS, rr, u[j] + ((ka.hhT != nullptr && t < T) ? ka.hhT[t] : ka.hh));
        }
        if (ka.mean != nullptr) {
            double ee[RAND ? SUB : 1][RAND ? D : 1];      // (RAND) the lane's transition draws: 8 D consecutive doubles of eps_t
            if constexpr (RAND) {
#pragma unroll
                for (int j = 0; j < SUB; ++j)
#pragma unroll
                    for (int k = 0; k < D; ++k) ee[j][k] = (t0 + j < T) ? ka.eps_t[(t0 + j) * D + k] : 0.0;
            }
#pragma unroll
            for (int j = SUB - 1; j >= 0; --j) {
                double o = o0[j];
#pragma unroll
                for (int k = 0; k < D; ++k) o = fma(ka.h[k], xi[k], o);
                o0[j] = o;
                double np[D];
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    double v = ka.c[i] * r[j];
#pragma unroll
                    for (int k = 0; k < D; ++k) v = fma(ka.G[i][k], xi[k], v);
                    if constexpr (RAND) {      // + (U' eps)_i; the series' last step also takes G xi_T
#pragma unroll
                        for (int k = 0; k <= i; ++k) v = fma(ka.U[k][i], ee[j][k], v);
                        if (t0 + j == T - 1) v += ka.v0[i];
                    }
                    np[i] = v;
                }
#pragma unroll
                for (int i = 0; i < D; ++i) xi[i] = np[i];
            }
            // reverse scan: lane l <- sum_{m >= l} G^(8 (m - l)) xi_m
#define TGP_SM_BWD(K)                                                                      \
    do {                                                                                   \
        double g_[D], n_[D];                                                               \
        _Pragma("unroll") for (int i = 0; i < D; ++i) {                                    \
            g_[i] = dpp_mov<0x100 + (1 << (K))>(xi[i]);                                    \
            n_[i] = xi[i];                                                                 \
        }                                                                                  \
        _Pragma("unroll") for (int i = 0; i < D; ++i)                                      \
            _Pragma("unroll") for (int k = 0; k < D; ++k) n_[i] = fma(ka.GP[K][i][k], g_[k], n_[i]); \
        _Pragma("unroll") for (int i = 0; i < D; ++i) xi[i] = n_[i];                       \
    } while (0)
            TGP_SM_BWD(0);
            TGP_SM_BWD(1);
            TGP_SM_BWD(2);
            TGP_SM_BWD(3);
#undef TGP_SM_BWD
            {
                const int e1 = 16 - (lane & 15);      // rows 0 and 2 take the first lane of the row above them, 16 - p lanes away
                double gv[D], nv[D];
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    gv[i] = dpp_mov<0x15F, 0x5>(dpp_mov<0x130>(xi[i]));
                    nv[i] = xi[i];
                }
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int k = 0; k < D; ++k) nv[i] = fma(sPg[i][k][e1], gv[k], nv[i]);
#pragma unroll
                for (int i = 0; i < D; ++i) xi[i] = nv[i];
                const int e2 = lane < 32 ? 32 - lane : 0;      // the lower half takes lane 32
                const double keep = lane < 32 ? 1.0 : 0.0;
#pragma unroll
                for (int i = 0; i < D; ++i) gv[i] = readlane_d(xi[i], 32) * keep;
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int k = 0; k < D; ++k) nv[i] = fma(sPg[i][k][e2], gv[k], nv[i]);
#pragma unroll
                for (int i = 0; i < D; ++i) xi[i] = nv[i];
            }
        }
    }
    if (ka.mean != nullptr) {      // (kernel-uniform)
        double pin[D];      // xi behind the lane's last step: its right neighbour's (lane 63: zero)
#pragma unroll
        for (int i = 0; i < D; ++i) pin[i] = dpp_mov<0x130>(xi[i]);
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < D; ++i) sB[wave][i] = any_valid ? xi[i] : 0.0;
        }
        __syncthreads();
        if (any_valid) {
            double zin[D];
#pragma unroll
            for (int i = 0; i < D; ++i) zin[i] = 0.0;
#pragma unroll
            for (int k = 1; k <= 3; ++k) {
                const int src = wave + k;
                if (src >= NW) continue;
                if (k == 1) {
#pragma unroll
                    for (int i = 0; i < D; ++i) zin[i] += sB[src][i];
                } else {
#pragma unroll
                    for (int i = 0; i < D; ++i)
#pragma unroll
                        for (int m = 0; m < D; ++m) zin[i] = fma(ka.GPT[k - 2][i][m], sB[src][m], zin[i]);
                }
            }
            const int eb = 63 - lane;
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int k = 0; k < D; ++k) pin[i] = fma(sPg[i][k][eb], zin[k], pin[i]);
            if (first && wave == 0 && lane == 0) {      // xi at step nhs: where the host's half takes over (G^512 zin + the tile's own)
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    double v = xi[i];
#pragma unroll
                    for (int k = 0; k < D; ++k) v = fma(ka.GPT[0][i][k], zin[k], v);
                    ka.xi_out[i] = v;
                }
                if (ka.xi_flag != nullptr) {
                    __threadfence_system();
                    __hip_atomic_store(ka.xi_flag, 2 * ka.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
            // ---- outputs of the steps the workgroup owns
            const bool whole = t0 >= c_lo && t0 + SUB <= c_hi;
            const bool al = ((reinterpret_cast<uintptr_t>(ka.mean) | (RAND ? (uintptr_t)0 : reinterpret_cast<uintptr_t>(ka.var))) & 15) == 0;
            const double rn0 = ka.rnew_per_step ? 0.0 : ka.Rnew[0];
            const bool in_tail = t0 + SUB > T - ka.n1;
            const bool wide = whole && al;      // (t0 is a multiple of 8: spans start on multiples of 16 behind nhs, itself one)
#pragma unroll
            for (int j2 = 0; j2 < SUB; j2 += 2) {
                double om[2], ov[2];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int j = j2 + jj;
                    const long long t = t0 + j;
                    double o = o0[j];
#pragma unroll
                    for (int k = 0; k < D; ++k) o = fma(ka.WG[j][k], pin[k], o);
                    double v = 0.0;
                    if constexpr (RAND) {      // the draw: + h' xi_T at the series' last step, + sqrt(Rnew_t) eta_t
                        if (t == T - 1) o += ka.s0;
                        if (t < T) o = fma(sqrt(ka.rnew_per_step ? ka.Rnew[t] : rn0), ka.eps_e[t], o);
                    } else {
                        v = ka.vb;
                        if (in_tail && t < T && T - 1 - t < ka.n1) v = ka.tvb[T - 1 - t];
                        if (ka.rnew_per_step) v += (t < T ? ka.Rnew[t] : 0.0);
                        else v += rn0;
                    }
                    om[jj] = o;
                    ov[jj] = v;
                }
                if (wide) {
                    v2d w;
                    w.x = om[0];
                    w.y = om[1];
                    reinterpret_cast<v2d*>(ka.mean + t0)[j2 / 2] = w;
                    if constexpr (!RAND) {
                        w.x = ov[0];
                        w.y = ov[1];
                        reinterpret_cast<v2d*>(ka.var + t0)[j2 / 2] = w;
                    }
                } else {
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const long long t = t0 + j2 + jj;
                        if (t >= c_lo && t < c_hi) {
                            ka.mean[t] = om[jj];
                            if constexpr (!RAND) ka.var[t] = ov[jj];
                        }
                    }
                }
            }
        }
    }
    if (ka.head_out_flag != nullptr && ka.mean != nullptr && g == ((ka.nwg + 7) / 8) / 2) {      // the head's outputs, by a workgroup from the middle of the dispatch (see k_steady_one)
        if (wave < (RAND ? 1 : 2)) {
            wait_tables(ka.head_out_flag, ka.seq, &sPoison);
            double* dst = wave == 0 ? ka.mean : ka.var;
            const double* src = ka.head_out + (wave == 0 ? 0 : ka.nhs);
            for (long long t = lane; t < ka.nhs; t += 64) dst[t] = src[t];
        }
    }
    acc = wave_sum_to_lane63(acc);
    if (lane == 63) sAcc[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tsum = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) tsum += sAcc[w];
        ka.part[g] = tsum + sPoison;
    }
}

template <int D, int NW>
int launch_smooth(hipStream_t st, const tgp_plan::SmoothPlan& sp, const double* mu0, const SmoothCall& c) {
    const tgp_plan::FilterPlan& fp = sp.fp;
    SArgs<D> ka;
    static_assert(sizeof(SArgs<D>) <= 11264, "the kernel-argument segment (12 KB launch on gfx950: scripts/micro/bigarg.hip)");
    std::memset(&ka, 0, sizeof ka);
    for (int i = 0; i < D; ++i) {
        ka.a[i] = fp.a[i];
        ka.kA[i] = fp.kA[i];
        ka.h[i] = fp.h[i];
        ka.c[i] = sp.c[i];
        ka.mu0[i] = mu0 != nullptr ? mu0[i] : 0.0;
        for (int j = 0; j < kWJ; ++j) {
            ka.WJ[j][i] = sp.WJ[j][i];
            ka.WG[j][i] = sp.WG[j][i];
        }
        for (int k = 0; k < D; ++k) {
            ka.Phi[i][k] = fp.Phi[i * D + k];
            ka.G[i][k] = sp.G[i * D + k];
            for (int b = 0; b < 6; ++b) {
                ka.P[b][i][k] = fp.P[b][i * D + k];
                ka.GP[b][i][k] = sp.GP[b][i * D + k];
            }
            for (int b = 0; b < 2; ++b) {
                ka.PT[b][i][k] = fp.PT[b][i * D + k];
                ka.GPT[b][i][k] = sp.GPT[b][i * D + k];
            }
        }
    }
    ka.hh = fp.hh;
    ka.rS = sp.rS;
    ka.vb = sp.vb;
    ka.T = c.T;
    ka.nhs = fp.nhs;
    ka.n1 = sp.n1;
    const bool post = c.mean != nullptr;
    ka.halo = sp.halo;
    ka.halo_r = post ? sp.halo : 0;
    ka.C = smooth_span(sp, post);
    ka.nwg = (c.T - fp.nhs + ka.C - 1) / ka.C;
    ka.rnew_per_step = c.rnew_per_step;
    ka.y = c.y;
    ka.Rnew = c.Rnew;
    ka.tvb = c.tvb;
    ka.mean = c.mean;
    ka.var = c.var;
    ka.part = c.part;
    ka.xi_out = c.xi_out;
    ka.hhT = c.hh_t;
    ka.mu0p = c.mu0;
    ka.mu0_flag = c.mu0_flag;
    ka.xi_flag = c.xi_flag;
    ka.head_in = c.head_in;
    ka.head_in_flag = c.head_in_flag;
    ka.head_out = c.head_out;
    ka.head_out_flag = c.head_out_flag;
    ka.seq = c.seq;
    const long long per = (ka.nwg + 7) / 8;
    static const int minw_env = [] { const char* v = std::getenv("TGP_SMOOTH_MINW"); return v ? std::atoi(v) : 0; }();
    const bool four = minw_env ? minw_env >= 4 : D <= 6;
    if (c.eps_t != nullptr) {      // a draw from the posterior (d <= kSmoothRandMaxD: the lane's 8 d draws sit in registers)
        if constexpr (D <= kSmoothRandMaxD) {
            ka.eps_t = c.eps_t;
            ka.eps_e = c.eps_e;
            for (int i = 0; i < D; ++i) {
                ka.v0[i] = c.v0[i];
                for (int k = 0; k < D; ++k) ka.U[i][k] = c.U[i * D + k];
            }
            ka.s0 = c.s0;
            hipLaunchKernelGGL((k_smooth_one<D, NW, 2, true>), dim3((unsigned)(per * 8)), dim3(NW * 64), 0, st, ka);
            return (int)hipGetLastError();
        } else {
            return (int)hipErrorInvalidValue;
        }
    }
    if (four && D <= 6) hipLaunchKernelGGL((k_smooth_one<D, NW, (D <= 6 ? 4 : 2), false>), dim3((unsigned)(per * 8)), dim3(NW * 64), 0, st, ka);
    else hipLaunchKernelGGL((k_smooth_one<D, NW, 2, false>), dim3((unsigned)(per * 8)), dim3(NW * 64), 0, st, ka);
    return (int)hipGetLastError();
}
}  // namespace

// (sixteen waves per workgroup -- spans of 8192 steps, half the halo overhead, the tables' rows dealt over sixteen waves -- were measured
//  SLOWER: fused call 0.228 against 0.177 ms at d = 6, T = 1e7: one workgroup per CU leaves nothing to run beside its barriers)
long long smooth_span(const tgp_plan::SmoothPlan& sp, bool post) { return 8LL * 64 * kWJ - (post ? 2LL : 1LL) * sp.halo; }
long long smooth_workgroups(const tgp_plan::SmoothPlan& sp, long long T, bool post) {
    const long long C = smooth_span(sp, post);
    return C > 0 ? (T - sp.fp.nhs + C - 1) / C : -1;
}

int smooth_lti(hipStream_t stream, const tgp_plan::SmoothPlan& sp, const double* mu_start, const SmoothCall& c) {
    switch (sp.fp.d) {
        case 1: return launch_smooth<1, 8>(stream, sp, mu_start, c);
        case 2: return launch_smooth<2, 8>(stream, sp, mu_start, c);
        case 3: return launch_smooth<3, 8>(stream, sp, mu_start, c);
        case 4: return launch_smooth<4, 8>(stream, sp, mu_start, c);
        case 5: return launch_smooth<5, 8>(stream, sp, mu_start, c);
        case 6: return launch_smooth<6, 8>(stream, sp, mu_start, c);
        case 7: return launch_smooth<7, 8>(stream, sp, mu_start, c);
        case 8: return launch_smooth<8, 8>(stream, sp, mu_start, c);
    }
    return (int)hipErrorInvalidValue;
}

bool overlap_allowed() { return overlap_tables(); }

void plan_smooth(const tgp_plan::ModelHost& m, long long T, tgp_plan::SmoothPlan& sp, double* tvb, bool post) {
    if (!host_cpu_ok()) {
        sp.why = tgp_plan::kEigFail;
        return;
    }
    tgp_plan::build_smooth_any(m, T, sp, tvb, post);
    // (halos would eat three quarters of a span.  Workgroup g >= 1 starts `halo` steps in front of its range; with a span shorter than a halo that
    //  reaches back into the head -- or in front of the series: a memory fault found by the randomised sweep, T = 250 000, two Matern-1/2 of one
    //  length scale, halo 1520 against spans of 1056.
    //  Such a workgroup -- only the second one can be: 2 spans + a halo fit its tiles exactly when a span is shorter than a halo -- starts behind
    //  the head like the first, from the head's exact end state (`from_head` in the kernels).
    if (sp.why == tgp_plan::kOk && smooth_span(sp, post) < 1024) sp.why = tgp_plan::kSlowMixing;
}
void plan_smooth_head_forward(const tgp_plan::ModelHost& m, const tgp_plan::SmoothPlan& sp, const double* y, double* mu_end, double* quad, const double* hh_t) {
    tgp_plan::smooth_head_forward_any(m, sp, y, mu_end, quad, hh_t);
}
bool plan_smooth_head_tables(const tgp_plan::ModelHost& m, const tgp_plan::SmoothPlan& sp) { return tgp_plan::smooth_head_tables_any(m, sp); }
void plan_smooth_head_backward(const tgp_plan::ModelHost& m, const tgp_plan::SmoothPlan& sp, const double* y, const double* lam, double* mean, double* vb) {
    tgp_plan::smooth_head_backward_any(m, sp, y, lam, mean, vb);
}
bool plan_smooth_rand_factors(const tgp_plan::SmoothPlan& sp, const double* eps0, double* U, double* v0, double* s0) {
    return tgp_plan::smooth_rand_factors_any(sp, eps0, U, v0, s0);
}
void plan_smooth_head_backward_rand(const tgp_plan::ModelHost& m, const tgp_plan::SmoothPlan& sp, const double* y, const double* delta, const double* eps_e,
                                    const double* eps_t, const double* rn, bool rn_per_step, double* out) {
    tgp_plan::smooth_head_backward_rand_any(m, sp, y, delta, eps_e, eps_t, rn, rn_per_step, out);
}

}  // namespace tgp_modal
