// Stationary-gain engine for wide states (8 < d <= 63) -- see tgp_wide.hpp.  gfx950 only (wave64).
#include "tgp_wide.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

#include "tgp_alloc.hpp"

namespace tgp_wide {

namespace {

typedef double v2d __attribute__((ext_vector_type(2)));

struct ZArg {
    double z[64];      // the filtered mean behind the head, one component per lane (zero beyond d)
};

__device__ __forceinline__ void lds_sync() {      // one wave talking to itself through LDS (DS operations of a wave execute in order)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ double readlane_d(double x, int l) {      // l wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), l), hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
    return __hiloint2double(hi, lo);
}

// tab: [DP + 2][64] -- column j of the lanes' rows (lane i < d: row i of Phi; lane d: -g; zero beyond), then the lanes' input gains (K_i; 1 for the
// observer) and constants (c_i; -g0 for the observer).  One wave per chunk [s0, s1) of the steps behind the head; its warm-up starts `halo` steps
// early from zero, or at the head's end from the head's own end state where that is nearer.
template <int DP>
__global__ __launch_bounds__(64) void k_wide_lml(const double* __restrict__ tab, const double* __restrict__ y, double hh, long long T, long long t_head,
                                                  long long chunk_len, long long halo, int obs_lane, ZArg z0, double* __restrict__ part, double* __restrict__ rout,
                                                  const double* __restrict__ ht, double* __restrict__ mout, int d) {
    __shared__ __attribute__((aligned(16))) double zb[64];
    const int lane = threadIdx.x;
    const long long chunk = blockIdx.x;
    const long long s0 = t_head + chunk * chunk_len;
    long long s1 = s0 + chunk_len;
    if (s1 > T) s1 = T;
    // (ht: an emission offset PER STEP -- a mean function at the inputs -- instead of the shared hh: the gains do not see it)
    auto obs = [&](long long t) { return y[t] - (ht != nullptr ? ht[t] : 0.0); };
    const bool from_head = s0 - halo <= t_head;
    const long long w0 = from_head ? t_head : s0 - halo;
    double phi[DP];
#pragma unroll
    for (int j = 0; j < DP; ++j) phi[j] = tab[(size_t)j * 64 + lane];
    const double kin = tab[(size_t)DP * 64 + lane], cin = tab[(size_t)(DP + 1) * 64 + lane];
    zb[lane] = from_head ? z0.z[lane] : 0.0;
    lds_sync();
    double ssq = 0.0;
    double yn = (w0 + lane < s1) ? obs(w0 + lane) : 0.0;
    for (long long tb = w0; tb < s1; tb += 64) {
        const double yv = yn;
        yn = (tb + 64 + lane < s1) ? obs(tb + 64 + lane) : 0.0;      // (the next block's observations: on their way while this block runs)
        const int nb = (int)((s1 - tb < 64) ? (s1 - tb) : 64);
        const bool keep = rout != nullptr && tb + 64 > s0;          // (a posterior call: the innovations of the chunk's own steps go to memory)
        double outr = 0.0;
        for (int l = 0; l < nb; ++l) {
            const double u = readlane_d(yv, l) - hh;
            double a0 = fma(kin, u, cin), a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
            for (int j = 0; j < DP; j += 8) {
                const v2d q0 = *reinterpret_cast<const v2d*>(&zb[j]), q1 = *reinterpret_cast<const v2d*>(&zb[j + 2]);
                const v2d q2 = *reinterpret_cast<const v2d*>(&zb[j + 4]), q3 = *reinterpret_cast<const v2d*>(&zb[j + 6]);
                a0 = fma(phi[j], q0.x, a0);
                a1 = fma(phi[j + 1], q0.y, a1);
                a2 = fma(phi[j + 2], q1.x, a2);
                a3 = fma(phi[j + 3], q1.y, a3);
                a0 = fma(phi[j + 4], q2.x, a0);
                a1 = fma(phi[j + 5], q2.y, a1);
                a2 = fma(phi[j + 6], q3.x, a2);
                a3 = fma(phi[j + 7], q3.y, a3);
            }
            const double acc = (a0 + a1) + (a2 + a3);
            lds_sync();      // (every lane has read the old state)
            zb[lane] = acc;
            lds_sync();
            if (tb + l >= s0) ssq = fma(acc, acc, ssq);      // (the observer's acc is the step's innovation)
            if (mout != nullptr && tb + l >= s0 && lane < d) mout[(tb + l) * d + lane] = acc;      // (_filter: the filtered mean of the chunk's own steps)
            if (keep) {
                const double rr = readlane_d(acc, obs_lane);
                outr = lane == l ? rr : outr;
            }
        }
        if (keep && lane < nb && tb + lane >= s0) rout[tb + lane] = outr;
    }
    const double s = readlane_d(ssq, obs_lane);
    if (lane == 0) part[chunk] = s;
}

// The backward half (Bryson-Frazier in the predicted form): lam_t = h r_t / S + Psi lam_(t+1), Psi = (I - h K') A';
// mean_t = y_t - (R / S) r_t + gw . lam_(t+1), gw = R A K; var_t = (S - R) R / S - gw' Lam_(t+1) gw + Rnew_t, where the quadratic form is a constant
// behind the last n1 steps (qtab: its partial sums at the series' end).  tab: [DP + 1][64] -- column j of the lanes' rows (lane i < d: row i of Psi;
// lane d, the observer: gw), then the gains on r_t (h_i / S; observer: -R / S).  A chunk starts `halo` steps behind its end from lam = 0 (exact at T).
template <int DP>
__global__ __launch_bounds__(64) void k_wide_bwd(const double* __restrict__ tab, const double* __restrict__ y, const double* __restrict__ r,
                                                  const double* __restrict__ Rnew, int rnew_per_step, const double* __restrict__ qtab, long long n1, double vbase,
                                                  double qinf, long long T, long long t_head, long long chunk_len, long long halo, int obs_lane, int d,
                                                  double* __restrict__ mean, double* __restrict__ var, double* __restrict__ lam_out) {
    __shared__ __attribute__((aligned(16))) double zb[64];
    const int lane = threadIdx.x;
    const long long chunk = blockIdx.x;
    const long long s0 = t_head + chunk * chunk_len;
    long long s1 = s0 + chunk_len;
    if (s1 > T) s1 = T;
    long long w1 = s1 + halo;
    if (w1 > T) w1 = T;
    double phi[DP];
#pragma unroll
    for (int j = 0; j < DP; ++j) phi[j] = tab[(size_t)j * 64 + lane];
    const double kin = tab[(size_t)DP * 64 + lane];
    zb[lane] = 0.0;
    lds_sync();
    double rn = (w1 - 1 - lane >= s0) ? r[w1 - 1 - lane] : 0.0;
    for (long long tb = w1 - 1; tb >= s0; tb -= 64) {      // the block holds the steps tb, tb - 1, ..., one per lane
        const double rv = rn;
        rn = (tb - 64 - lane >= s0) ? r[tb - 64 - lane] : 0.0;
        const int nb = (int)((tb - s0 + 1 < 64) ? (tb - s0 + 1) : 64);
        const bool own = tb - 63 < s1;      // (some step of the block is the chunk's own)
        double outm = 0.0;
        for (int l = 0; l < nb; ++l) {
            const double rr = readlane_d(rv, l);
            double a0 = kin * rr, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
            for (int j = 0; j < DP; j += 8) {
                const v2d q0 = *reinterpret_cast<const v2d*>(&zb[j]), q1 = *reinterpret_cast<const v2d*>(&zb[j + 2]);
                const v2d q2 = *reinterpret_cast<const v2d*>(&zb[j + 4]), q3 = *reinterpret_cast<const v2d*>(&zb[j + 6]);
                a0 = fma(phi[j], q0.x, a0);
                a1 = fma(phi[j + 1], q0.y, a1);
                a2 = fma(phi[j + 2], q1.x, a2);
                a3 = fma(phi[j + 3], q1.y, a3);
                a0 = fma(phi[j + 4], q2.x, a0);
                a1 = fma(phi[j + 5], q2.y, a1);
                a2 = fma(phi[j + 6], q3.x, a2);
                a3 = fma(phi[j + 7], q3.y, a3);
            }
            const double acc = (a0 + a1) + (a2 + a3);
            lds_sync();
            zb[lane] = acc;
            lds_sync();
            if (own) {
                const double mm = readlane_d(acc, obs_lane);
                outm = lane == l ? mm : outm;
            }
        }
        const long long t = tb - lane;
        if (own && lane < nb && t < s1) {
            mean[t] = y[t] + outm;
            const long long jt = T - 1 - t;
            const double q = jt < n1 ? qtab[jt] : qinf;
            var[t] = vbase - q + (rnew_per_step ? Rnew[t] : Rnew[0]);
        }
    }
    if (chunk == 0 && lane < d) lam_out[lane] = zb[lane];      // lam at the head's end: the head's backward pass runs on the host
}

// ---- d <= 31: FOUR chunks per wave, no LDS.  A row of sixteen lanes holds one chunk's state in two registers (lane p: components p and p + 16) and the
// two matching rows of the matrix; component j reaches the row's lanes as the DPP operand of the multiply-add itself (v_fmac_f64_dpp row_newbcast:j), and
// so does the step's observation out of the row's block of sixteen.  66 multiply-adds per step and wave for four chunks (the LDS form: 32 and 16 broadcast
// reads for one).  The table is the LDS kernels' (DP = 32).
// (measured: 8-9 cycles per v_fmac_f64_dpp at one wave per SIMD, four accumulator chains or two alike -- 0.187 ms at d = 28, T = 1e6; two 32-bit
//  DPP moves and two plain multiply-adds per component instead run at the full issue rate and come to 0.208 ms; the LDS form 0.36 ms)
template <int J>
__device__ __forceinline__ void fmac_bc(double& acc, double src, double mul) {      // acc += (lane J of the row's src) * mul
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(J));
}
struct Acc4 {
    double v[4];
    __device__ __forceinline__ double sum() const { return (v[0] + v[1]) + (v[2] + v[3]); }
};
template <int OFF, int... Js>
__device__ __forceinline__ void dot16(Acc4& a, double z, const double (&phi)[32], std::integer_sequence<int, Js...>) {
    (fmac_bc<Js>(a.v[Js % 4], z, phi[OFF + Js]), ...);
}
template <int OFF, int... Js>      // both outputs of the lane against the same sixteen components, their chains interleaved (eight independent accumulators)
__device__ __forceinline__ void dot16ab(Acc4& a, Acc4& b, double z, const double (&pA)[32], const double (&pB)[32], std::integer_sequence<int, Js...>) {
    ((fmac_bc<Js>(a.v[Js % 4], z, pA[OFF + Js]), fmac_bc<Js>(b.v[Js % 4], z, pB[OFF + Js])), ...);
}
typedef std::make_integer_sequence<int, 16> Seq16;

struct RowGeom {      // per lane, the same within a row
    long long s0, s1, w;      // own steps [s0, s1); w: where the recursion starts (forward: w <= s0, upwards; backward: w >= s1, downwards from w - 1)
    bool valid;
};

template <int L, bool KEEP, int NB>
__device__ __forceinline__ void fwd_step4(double& yv, double& zlo, double& zhi, const double (&pA)[32], const double (&pB)[32], double kA, double kB, double cA, double cB,
                                          long long t, const RowGeom& g, bool obsB, bool is_obs, double& ssq, double* __restrict__ rout, double* __restrict__ mout, int d, int p) {
    // (a DPP operand must not be read within two cycles of the VALU write of its register, nor within five of a write to EXEC: the recogniser that
    //  spaces such pairs does not look into inline assembly)
    asm volatile("s_nop 4" : "+v"(zlo), "+v"(zhi), "+v"(yv) : : "memory");
    Acc4 a{{cA, 0.0, 0.0, 0.0}}, b{{cB, 0.0, 0.0, 0.0}};
    fmac_bc<L>(a.v[3], yv, kA);
    if constexpr (NB == 2) {
        fmac_bc<L>(b.v[3], yv, kB);
        dot16ab<0>(a, b, zlo, pA, pB, Seq16{});
        dot16ab<16>(a, b, zhi, pA, pB, Seq16{});
    } else {      // (d <= 15: one component per lane, sixteen columns)
        dot16<0>(a, zlo, pA, Seq16{});
    }
    const double nA = a.sum(), nB = NB == 2 ? b.sum() : 0.0;
    const bool live = t < g.s1;
    zlo = live ? nA : zlo;
    zhi = live ? nB : zhi;
    const bool own = live && t >= g.s0;
    double rr = obsB ? nB : nA;
    rr = own ? rr : 0.0;
    ssq = fma(rr, rr, ssq);
    if (KEEP) {
        if (is_obs && own) rout[t] = rr;
    }
    if (mout != nullptr) {      // (_filter: the filtered mean of the chunk's own steps -- wave-uniform test)
        if (own && p < d) mout[t * d + p] = nA;
        if (NB == 2 && own && 16 + p < d) mout[t * d + 16 + p] = nB;
    }
}
template <bool KEEP, int NB, int... Ls>
__device__ __forceinline__ void fwd_block4(double& yv, double& zlo, double& zhi, const double (&pA)[32], const double (&pB)[32], double kA, double kB, double cA, double cB,
                                           long long t0, const RowGeom& g, bool obsB, bool is_obs, double& ssq, double* __restrict__ rout, double* __restrict__ mout, int d, int p,
                                           std::integer_sequence<int, Ls...>) {
    (fwd_step4<Ls, KEEP, NB>(yv, zlo, zhi, pA, pB, kA, kB, cA, cB, t0 + Ls, g, obsB, is_obs, ssq, rout, mout, d, p), ...);
}

template <bool KEEP, int NB>
__global__ __launch_bounds__(64) void k_wide_lml4(const double* __restrict__ tab, const double* __restrict__ y, double hh, long long T, long long t_head, long long chunk_len,
                                                   long long halo, long long chunks, int d, ZArg z0, double* __restrict__ part, double* __restrict__ rout,
                                                   const double* __restrict__ ht, double* __restrict__ mout) {
    auto obs = [&](long long t) { return y[t] - (ht != nullptr ? ht[t] : 0.0); };      // (see k_wide_lml)
    const int lane = threadIdx.x, p = lane & 15, row = lane >> 4;
    const long long chunk = (long long)blockIdx.x * 4 + row;
    RowGeom g;
    g.valid = chunk < chunks;
    g.s0 = t_head + chunk * chunk_len;
    g.s1 = g.s0 + chunk_len;
    if (g.s1 > T) g.s1 = T;
    const bool from_head = g.s0 - halo <= t_head;
    g.w = from_head ? t_head : g.s0 - halo;
    if (!g.valid) g.s0 = g.s1 = g.w = T;
    double pA[32], pB[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        pA[j] = j < 16 * NB ? tab[(size_t)j * 64 + p] : 0.0;
        pB[j] = NB == 2 ? tab[(size_t)j * 64 + 16 + p] : 0.0;
    }
    const double kA = tab[(size_t)32 * 64 + p], kB = tab[(size_t)32 * 64 + 16 + p];
    const double cA = tab[(size_t)33 * 64 + p] - kA * hh, cB = tab[(size_t)33 * 64 + 16 + p] - kB * hh;      // (u = y - hh folded into the constant)
    double zlo = (g.valid && from_head) ? z0.z[p] : 0.0, zhi = (g.valid && from_head) ? z0.z[16 + p] : 0.0;
    const bool obsB = d >= 16, is_obs = p == (d & 15);
    // the longest row's number of steps (wave-uniform)
    const long long len = g.s1 - g.w;
    long long nmax = 0;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        const int lo = __builtin_amdgcn_readlane((int)(len & 0xffffffffll), 16 * r4), hi = __builtin_amdgcn_readlane((int)(len >> 32), 16 * r4);
        const long long v = ((long long)hi << 32) | (unsigned)lo;
        nmax = v > nmax ? v : nmax;
    }
    double ssq = 0.0;
    double yn = (g.w + p < g.s1) ? obs(g.w + p) : 0.0;
    for (long long kb = 0; kb < nmax; kb += 16) {
        double yv = yn;
        yn = (g.w + kb + 16 + p < g.s1) ? obs(g.w + kb + 16 + p) : 0.0;      // (the next block: on its way while this one runs)
        fwd_block4<KEEP, NB>(yv, zlo, zhi, pA, pB, kA, kB, cA, cB, g.w + kb, g, obsB, is_obs, ssq, rout, mout, d, p, Seq16{});
    }
    if (is_obs && g.valid) part[chunk] = ssq;
}

template <int L, int NB>
__device__ __forceinline__ void bwd_step4(double& rv, double& yv, double& zlo, double& zhi, const double (&pA)[32], const double (&pB)[32], double kA, double kB, double yA, double yB,
                                          long long t, const RowGeom& g, bool obsB, bool is_obs, double* __restrict__ mean) {
    asm volatile("s_nop 4" : "+v"(zlo), "+v"(zhi), "+v"(rv), "+v"(yv) : : "memory");
    Acc4 a{{0.0, 0.0, 0.0, 0.0}}, b{{0.0, 0.0, 0.0, 0.0}};
    fmac_bc<L>(a.v[3], rv, kA);
    fmac_bc<L>(a.v[2], yv, yA);      // (1 at the observer: its sum is the step's mean; its slot of the state multiplies a zero column)
    if constexpr (NB == 2) {
        fmac_bc<L>(b.v[3], rv, kB);
        fmac_bc<L>(b.v[2], yv, yB);
        dot16ab<0>(a, b, zlo, pA, pB, Seq16{});
        dot16ab<16>(a, b, zhi, pA, pB, Seq16{});
    } else {
        dot16<0>(a, zlo, pA, Seq16{});
    }
    const double nA = a.sum(), nB = NB == 2 ? b.sum() : 0.0;
    const bool live = t >= g.s0;
    zlo = live ? nA : zlo;
    zhi = live ? nB : zhi;
    if (is_obs && live && t < g.s1) mean[t] = obsB ? nB : nA;
}
template <int NB, int... Ls>
__device__ __forceinline__ void bwd_block4(double& rv, double& yv, double& zlo, double& zhi, const double (&pA)[32], const double (&pB)[32], double kA, double kB, double yA, double yB,
                                           long long t0, const RowGeom& g, bool obsB, bool is_obs, double* __restrict__ mean, std::integer_sequence<int, Ls...>) {
    (bwd_step4<Ls, NB>(rv, yv, zlo, zhi, pA, pB, kA, kB, yA, yB, t0 - Ls, g, obsB, is_obs, mean), ...);
}

template <int NB>
__global__ __launch_bounds__(64) void k_wide_bwd4(const double* __restrict__ tab, const double* __restrict__ y, const double* __restrict__ r, const double* __restrict__ Rnew,
                                                   int rnew_per_step, const double* __restrict__ qtab, long long n1, double vbase, double qinf, long long T, long long t_head,
                                                   long long chunk_len, long long halo, long long chunks, int d, double* __restrict__ mean, double* __restrict__ var,
                                                   double* __restrict__ lam_out) {
    const int lane = threadIdx.x, p = lane & 15, row = lane >> 4;
    const long long chunk = (long long)blockIdx.x * 4 + row;
    RowGeom g;
    g.valid = chunk < chunks;
    g.s0 = t_head + chunk * chunk_len;
    g.s1 = g.s0 + chunk_len;
    if (g.s1 > T) g.s1 = T;
    g.w = g.s1 + halo;
    if (g.w > T) g.w = T;
    if (!g.valid) g.s0 = g.s1 = g.w = T;
    double pA[32], pB[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        pA[j] = j < 16 * NB ? tab[(size_t)j * 64 + p] : 0.0;
        pB[j] = NB == 2 ? tab[(size_t)j * 64 + 16 + p] : 0.0;
    }
    const double kA = tab[(size_t)32 * 64 + p], kB = tab[(size_t)32 * 64 + 16 + p];
    const bool obsB = d >= 16, is_obs = p == (d & 15);
    const double yA = (is_obs && !obsB) ? 1.0 : 0.0, yB = (is_obs && obsB) ? 1.0 : 0.0;
    double zlo = 0.0, zhi = 0.0;
    const long long len = g.w - g.s0;
    long long nmax = 0;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        const int lo = __builtin_amdgcn_readlane((int)(len & 0xffffffffll), 16 * r4), hi = __builtin_amdgcn_readlane((int)(len >> 32), 16 * r4);
        const long long v = ((long long)hi << 32) | (unsigned)lo;
        nmax = v > nmax ? v : nmax;
    }
    const long long top = g.w - 1;      // the row's first step
    double rn = (top - p >= g.s0) ? r[top - p] : 0.0;
    for (long long kb = 0; kb < nmax; kb += 16) {      // the block holds the steps top - kb, top - kb - 1, ..., one per lane of the row
        double rv = rn;
        const long long tl = top - kb - p;
        rn = (tl - 16 >= g.s0) ? r[tl - 16] : 0.0;
        const bool mine = tl >= g.s0 && tl < g.s1;
        double yv = mine ? y[tl] : 0.0;
        bwd_block4<NB>(rv, yv, zlo, zhi, pA, pB, kA, kB, yA, yB, top - kb, g, obsB, is_obs, mean, Seq16{});
        if (mine) {
            const long long jt = T - 1 - tl;
            const double q = jt < n1 ? qtab[jt] : qinf;
            var[tl] = vbase - q + (rnew_per_step ? Rnew[tl] : Rnew[0]);
        }
    }
    if (g.valid && chunk == 0) {      // lam at the head's end: the head's backward pass runs on the host
        if (p < d) lam_out[p] = zlo;
        if (16 + p < d) lam_out[16 + p] = zhi;
    }
}

// ---- rand (lgssm.jl:65-91 with the draws supplied): x_t = A x_(t-1) + a + U' eps_t (U = chol(Q + 1e-9 I), lgc.jl:84-87), y_t = h . x_t + hh + sqrt(R) e_t
// (lgc.jl:241-243).  The same shape as k_wide_lml -- a lane per component, the state round an LDS line -- with a second line for the step's draws; the
// open loop A forgets a state as the closed loop does, so a chunk warms up `halo` steps early from zero ON THE SAME DRAWS.  tab: [2 DP + 1][64] -- columns
// of the lanes' rows of A (observer, lane d: g = A' h), of U' (observer: U h), then the constants (a_i; observer: h . a + hh).
template <int DP>
__global__ __launch_bounds__(64) void k_wide_rand(const double* __restrict__ tab, const double* __restrict__ eps_t, const double* __restrict__ eps_e, double sqrtR,
                                                   long long T, long long chunk_len, long long halo, int obs_lane, int d, ZArg x0, double* __restrict__ y_out) {
    __shared__ __attribute__((aligned(16))) double zb[64];
    __shared__ __attribute__((aligned(16))) double eb[4][64];
    const int lane = threadIdx.x;
    const long long chunk = blockIdx.x;
    const long long s0 = chunk * chunk_len;
    long long s1 = s0 + chunk_len;
    if (s1 > T) s1 = T;
    const bool from_start = s0 - halo <= 0;
    const long long w0 = from_start ? 0 : s0 - halo;
    double pa[DP], pu[DP];
#pragma unroll
    for (int j = 0; j < DP; ++j) {
        pa[j] = tab[(size_t)j * 64 + lane];
        pu[j] = tab[(size_t)(DP + j) * 64 + lane];
    }
    const double cin = tab[(size_t)(2 * DP) * 64 + lane];
    zb[lane] = from_start ? x0.z[lane] : 0.0;
    lds_sync();
    auto draw = [&](long long t) { return (lane < d && t < s1) ? eps_t[t * d + lane] : 0.0; };
    // the draws of four steps ahead are on their way while a step runs
    double e0 = draw(w0), e1 = draw(w0 + 1), e2 = draw(w0 + 2), e3 = draw(w0 + 3);
    double outy = 0.0;
    for (long long t4 = w0; t4 < s1; t4 += 4) {
        eb[0][lane] = e0;
        eb[1][lane] = e1;
        eb[2][lane] = e2;
        eb[3][lane] = e3;
        e0 = draw(t4 + 4);
        e1 = draw(t4 + 5);
        e2 = draw(t4 + 6);
        e3 = draw(t4 + 7);
        lds_sync();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long t = t4 + k;
            if (t < s1) {      // (wave-uniform)
                double a0 = cin, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
                for (int j = 0; j < DP; j += 4) {
                    const v2d q0 = *reinterpret_cast<const v2d*>(&zb[j]), q1 = *reinterpret_cast<const v2d*>(&zb[j + 2]);
                    const v2d r0 = *reinterpret_cast<const v2d*>(&eb[k][j]), r1 = *reinterpret_cast<const v2d*>(&eb[k][j + 2]);
                    a0 = fma(pa[j], q0.x, a0);
                    a1 = fma(pa[j + 1], q0.y, a1);
                    a2 = fma(pa[j + 2], q1.x, a2);
                    a3 = fma(pa[j + 3], q1.y, a3);
                    a0 = fma(pu[j], r0.x, a0);
                    a1 = fma(pu[j + 1], r0.y, a1);
                    a2 = fma(pu[j + 2], r1.x, a2);
                    a3 = fma(pu[j + 3], r1.y, a3);
                }
                const double acc = (a0 + a1) + (a2 + a3);
                lds_sync();
                zb[lane] = acc;
                lds_sync();
                if (t >= s0) {
                    const double yy = readlane_d(acc, obs_lane);
                    outy = lane == (int)((t - s0) & 63) ? yy : outy;
                    if (((t - s0) & 63) == 63 || t == s1 - 1) {      // (a block of up to 64 emissions: one coalesced store)
                        const long long tb = t - ((t - s0) & 63), tl = tb + lane;
                        if (tl <= t) y_out[tl] = outy + sqrtR * eps_e[tl];
                    }
                }
            }
        }
    }
}

// ---- 32 <= d <= 47: the same four chunks per wave with THREE components per lane (p, p + 16, p + 32) and three rows of the matrix of 48 columns each --
// 144 doubles of rows, a fifth of them in accumulation registers (v_accvgpr reads ahead of their multiply-adds); 146 multiply-adds per step and wave
// against the LDS form's 64 and 32 broadcast reads for ONE chunk.  The table is the LDS kernels' (DP = 64).
template <int OFF, int W, int... Js>
__device__ __forceinline__ void dot16w(double (&a)[4], double z, const double (&phi)[W], std::integer_sequence<int, Js...>) {
    (fmac_bc<Js>(a[Js % 4], z, phi[OFF + Js]), ...);
}
template <int L, bool KEEP>
__device__ __forceinline__ void fwd_step43(double& yv, double (&z)[3], const double (&P)[3][48], const double (&kin)[3], const double (&cin)[3], long long t, const RowGeom& g,
                                           int obs_o, bool is_obs, double& ssq, double* __restrict__ rout, double* __restrict__ mout, int d, int p) {
    asm volatile("s_nop 4" : "+v"(z[0]), "+v"(z[1]), "+v"(z[2]), "+v"(yv) : : "memory");      // (DPP hazards: see fwd_step4)
    double n[3];
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        double a[4] = {cin[o], 0.0, 0.0, 0.0};
        fmac_bc<L>(a[3], yv, kin[o]);
        dot16w<0, 48>(a, z[0], P[o], Seq16{});
        dot16w<16, 48>(a, z[1], P[o], Seq16{});
        dot16w<32, 48>(a, z[2], P[o], Seq16{});
        n[o] = (a[0] + a[1]) + (a[2] + a[3]);
    }
    const bool live = t < g.s1;
#pragma unroll
    for (int o = 0; o < 3; ++o) z[o] = live ? n[o] : z[o];
    const bool own = live && t >= g.s0;
    double rr = obs_o == 0 ? n[0] : (obs_o == 1 ? n[1] : n[2]);
    rr = own ? rr : 0.0;
    ssq = fma(rr, rr, ssq);
    if (KEEP) {
        if (is_obs && own) rout[t] = rr;
    }
    if (mout != nullptr) {
#pragma unroll
        for (int o = 0; o < 3; ++o)
            if (own && 16 * o + p < d) mout[t * d + 16 * o + p] = n[o];
    }
}
template <bool KEEP, int... Ls>
__device__ __forceinline__ void fwd_block43(double& yv, double (&z)[3], const double (&P)[3][48], const double (&kin)[3], const double (&cin)[3], long long t0, const RowGeom& g,
                                            int obs_o, bool is_obs, double& ssq, double* __restrict__ rout, double* __restrict__ mout, int d, int p, std::integer_sequence<int, Ls...>) {
    (fwd_step43<Ls, KEEP>(yv, z, P, kin, cin, t0 + Ls, g, obs_o, is_obs, ssq, rout, mout, d, p), ...);
}
template <bool KEEP>
__global__ __launch_bounds__(64) void k_wide_lml43(const double* __restrict__ tab, const double* __restrict__ y, double hh, long long T, long long t_head, long long chunk_len,
                                                    long long halo, long long chunks, int d, ZArg z0, double* __restrict__ part, double* __restrict__ rout,
                                                    const double* __restrict__ ht, double* __restrict__ mout) {
    auto obs = [&](long long t) { return y[t] - (ht != nullptr ? ht[t] : 0.0); };
    const int lane = threadIdx.x, p = lane & 15, row = lane >> 4;
    const long long chunk = (long long)blockIdx.x * 4 + row;
    RowGeom g;
    g.valid = chunk < chunks;
    g.s0 = t_head + chunk * chunk_len;
    g.s1 = g.s0 + chunk_len;
    if (g.s1 > T) g.s1 = T;
    const bool from_head = g.s0 - halo <= t_head;
    g.w = from_head ? t_head : g.s0 - halo;
    if (!g.valid) g.s0 = g.s1 = g.w = T;
    double P[3][48], kin[3], cin[3], z[3];
#pragma unroll
    for (int o = 0; o < 3; ++o) {
#pragma unroll
        for (int j = 0; j < 48; ++j) P[o][j] = tab[(size_t)j * 64 + 16 * o + p];
        kin[o] = tab[(size_t)64 * 64 + 16 * o + p];
        cin[o] = tab[(size_t)65 * 64 + 16 * o + p] - kin[o] * hh;
        z[o] = (g.valid && from_head) ? z0.z[16 * o + p] : 0.0;
    }
    const int obs_o = d >> 4;
    const bool is_obs = p == (d & 15);
    const long long len = g.s1 - g.w;
    long long nmax = 0;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        const int lo = __builtin_amdgcn_readlane((int)(len & 0xffffffffll), 16 * r4), hi = __builtin_amdgcn_readlane((int)(len >> 32), 16 * r4);
        const long long v = ((long long)hi << 32) | (unsigned)lo;
        nmax = v > nmax ? v : nmax;
    }
    double ssq = 0.0;
    double yn = (g.w + p < g.s1) ? obs(g.w + p) : 0.0;
    for (long long kb = 0; kb < nmax; kb += 16) {
        double yv = yn;
        yn = (g.w + kb + 16 + p < g.s1) ? obs(g.w + kb + 16 + p) : 0.0;
        fwd_block43<KEEP>(yv, z, P, kin, cin, g.w + kb, g, obs_o, is_obs, ssq, rout, mout, d, p, Seq16{});
    }
    if (is_obs && g.valid) part[chunk] = ssq;
}

template <int L>
__device__ __forceinline__ void bwd_step43(double& rv, double& yv, double (&z)[3], const double (&P)[3][48], const double (&kin)[3], const double (&yin)[3], long long t,
                                           const RowGeom& g, int obs_o, bool is_obs, double* __restrict__ mean) {
    asm volatile("s_nop 4" : "+v"(z[0]), "+v"(z[1]), "+v"(z[2]), "+v"(rv), "+v"(yv) : : "memory");
    double n[3];
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        double a[4] = {0.0, 0.0, 0.0, 0.0};
        fmac_bc<L>(a[3], rv, kin[o]);
        fmac_bc<L>(a[2], yv, yin[o]);      // (1 at the observer)
        dot16w<0, 48>(a, z[0], P[o], Seq16{});
        dot16w<16, 48>(a, z[1], P[o], Seq16{});
        dot16w<32, 48>(a, z[2], P[o], Seq16{});
        n[o] = (a[0] + a[1]) + (a[2] + a[3]);
    }
    const bool live = t >= g.s0;
#pragma unroll
    for (int o = 0; o < 3; ++o) z[o] = live ? n[o] : z[o];
    if (is_obs && live && t < g.s1) mean[t] = obs_o == 0 ? n[0] : (obs_o == 1 ? n[1] : n[2]);
}
template <int... Ls>
__device__ __forceinline__ void bwd_block43(double& rv, double& yv, double (&z)[3], const double (&P)[3][48], const double (&kin)[3], const double (&yin)[3], long long t0,
                                            const RowGeom& g, int obs_o, bool is_obs, double* __restrict__ mean, std::integer_sequence<int, Ls...>) {
    (bwd_step43<Ls>(rv, yv, z, P, kin, yin, t0 - Ls, g, obs_o, is_obs, mean), ...);
}
__global__ __launch_bounds__(64) void k_wide_bwd43(const double* __restrict__ tab, const double* __restrict__ y, const double* __restrict__ r, const double* __restrict__ Rnew,
                                                    int rnew_per_step, const double* __restrict__ qtab, long long n1, double vbase, double qinf, long long T, long long t_head,
                                                    long long chunk_len, long long halo, long long chunks, int d, double* __restrict__ mean, double* __restrict__ var,
                                                    double* __restrict__ lam_out) {
    const int lane = threadIdx.x, p = lane & 15, row = lane >> 4;
    const long long chunk = (long long)blockIdx.x * 4 + row;
    RowGeom g;
    g.valid = chunk < chunks;
    g.s0 = t_head + chunk * chunk_len;
    g.s1 = g.s0 + chunk_len;
    if (g.s1 > T) g.s1 = T;
    g.w = g.s1 + halo;
    if (g.w > T) g.w = T;
    if (!g.valid) g.s0 = g.s1 = g.w = T;
    const int obs_o = d >> 4;
    const bool is_obs = p == (d & 15);
    double P[3][48], kin[3], yin[3], z[3];
#pragma unroll
    for (int o = 0; o < 3; ++o) {
#pragma unroll
        for (int j = 0; j < 48; ++j) P[o][j] = tab[(size_t)j * 64 + 16 * o + p];
        kin[o] = tab[(size_t)64 * 64 + 16 * o + p];
        yin[o] = (is_obs && o == obs_o) ? 1.0 : 0.0;
        z[o] = 0.0;
    }
    const long long len = g.w - g.s0;
    long long nmax = 0;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        const int lo = __builtin_amdgcn_readlane((int)(len & 0xffffffffll), 16 * r4), hi = __builtin_amdgcn_readlane((int)(len >> 32), 16 * r4);
        const long long v = ((long long)hi << 32) | (unsigned)lo;
        nmax = v > nmax ? v : nmax;
    }
    const long long top = g.w - 1;
    double rn = (top - p >= g.s0) ? r[top - p] : 0.0;
    for (long long kb = 0; kb < nmax; kb += 16) {
        double rv = rn;
        const long long tl = top - kb - p;
        rn = (tl - 16 >= g.s0) ? r[tl - 16] : 0.0;
        const bool mine = tl >= g.s0 && tl < g.s1;
        double yv = mine ? y[tl] : 0.0;
        bwd_block43(rv, yv, z, P, kin, yin, top - kb, g, obs_o, is_obs, mean, Seq16{});
        if (mine) {
            const long long jt = T - 1 - tl;
            const double q = jt < n1 ? qtab[jt] : qinf;
            var[tl] = vbase - q + (rnew_per_step ? Rnew[tl] : Rnew[0]);
        }
    }
    if (g.valid && chunk == 0) {
#pragma unroll
        for (int o = 0; o < 3; ++o)
            if (16 * o + p < d) lam_out[16 * o + p] = z[o];
    }
}

// _filter's covariances behind the head: the settled one (block n0 - 1, which the head's copy has just put there) into every later block
__global__ __launch_bounds__(256) void k_wide_fill_cov(double* __restrict__ P, long long n0, long long T, int dd) {
    const double* __restrict__ src = P + (n0 - 1) * dd;
    const long long n = (T - n0) * dd;
    double* __restrict__ dst = P + n0 * dd;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) dst[i] = src[i % dd];
}

// ---- host: small dense linear algebra, row-major ---------------------------------------------------------------------------------------
void matmul(int d, const double* X, const double* Y, double* Z) {      // Z = X Y
    for (int i = 0; i < d; ++i) {
        double* zi = Z + (size_t)i * d;
        for (int j = 0; j < d; ++j) zi[j] = 0.0;
        for (int k = 0; k < d; ++k) {
            const double x = X[(size_t)i * d + k];
            const double* yk = Y + (size_t)k * d;
            for (int j = 0; j < d; ++j) zi[j] += x * yk[j];
        }
    }
}
double norm_inf(int d, const double* X) {
    double n = 0.0;
    for (int i = 0; i < d; ++i) {
        double s = 0.0;
        for (int j = 0; j < d; ++j) s += std::fabs(X[(size_t)i * d + j]);
        n = std::max(n, s);
    }
    return n;
}

}  // namespace

struct Engine {
    Info info{};
    bool have = false;
    std::vector<double> key;
    long long key_T = -1;
    int d = 0, dp = 0;
    std::vector<double> A, avec, hvec;      // row-major A, a, h
    double hh = 0.0, R = 0.0, g0 = 0.0, Sss = 0.0, sum_logS_head = 0.0;
    std::vector<double> x0m;
    std::vector<double> Kt, St;            // the head's gains [n0][d] and innovation variances [n0]
    std::vector<double> tab_host;          // the forward kernel's table (see k_wide_lml)
    // the posterior half of the plan (built by the first posterior call of a model)
    bool post_ready = false;
    int post_why = kOk;
    std::vector<double> tabb_host;         // the backward kernel's table (see k_wide_bwd)
    std::vector<double> qtab;              // partial sums of the variance's quadratic form at the series' end [n1 + 1]
    std::vector<double> headvar;           // (S_t - R) R / S_t - gw_t' Lam_(t+1) gw_t for the head's steps [n0]
    double vbase = 0.0, qinf = 0.0;
    double* dev = nullptr;                 // device: forward table | backward table | qtab
    size_t dev_cap = 0;
    bool dev_current = false, dev_post_current = false;
    double* rbuf = nullptr;                // device: the innovations of the steps behind the head [T]
    size_t rbuf_cap = 0;
    double* pinned = nullptr;              // [kHeadMax] head y | [kHeadMax] head Rnew | [kHeadMax] head means | [kHeadMax] head vars | [64] lam | [kMaxChunks] sums
    std::vector<double> head_r, head_m;
    const char* kname = "k_wide_lml<32>";
    std::vector<double> Pf_head;          // the head's filtered covariances [n0][d d] (row-major = column-major: symmetric), for _filter; empty: too large
    double* rand_dev = nullptr;           // device: k_wide_rand's table
    size_t rand_cap = 0;
};

namespace {
bool dpp_enabled() {      // TGP_WIDE_DPP=0: the LDS kernels for every d (A/B runs)
    static const bool on = [] {
        const char* sv = std::getenv("TGP_WIDE_DPP");
        return !(sv && sv[0] == '0');
    }();
    return on;
}
}  // namespace
Engine* create() { return new Engine(); }
void destroy(Engine* e) {
    if (!e) return;
    if (e->dev) (void)tgp_alloc::dev_free(e->dev);
    if (e->rbuf) (void)tgp_alloc::dev_free(e->rbuf);
    if (e->rand_dev) (void)tgp_alloc::dev_free(e->rand_dev);
    if (e->pinned) (void)tgp_alloc::host_free(e->pinned);
    delete e;
}
const Info& last_plan(const Engine* e) { return e->info; }
const char* kernel_name(const Engine* e) { return e->kname; }
void stationary(const Engine* e, double* K, double* S, double* vbase, double* qinf) {
    const int d = e->d, n0 = e->info.n0;
    if (K && n0 > 0)
        for (int i = 0; i < d; ++i) K[i] = e->Kt[(size_t)(n0 - 1) * d + i];
    if (S) *S = e->Sss;
    if (vbase) *vbase = e->vbase;
    if (qinf) *qinf = e->qinf;
}
bool filter_ready(const Engine* e) { return e->have && e->info.why == kOk && e->Pf_head.size() == (size_t)e->info.n0 * e->d * e->d; }

namespace {
template <class F>
void model_words(const ModelHost& m, F&& f) {
    const size_t d = (size_t)m.d;
    f(m.A, d * d); f(m.a, d); f(m.Q, d * d); f(m.H, d); f(&m.hh, 1); f(&m.R, 1); f(m.x0m, d); f(m.x0P, d * d);
}
bool same_model(const Engine* e, const ModelHost& m, long long T) {
    if (!e->have || e->key_T != T || e->d != m.d) return false;
    const double* k = e->key.data();
    bool same = true;
    model_words(m, [&](const double* p, size_t n) {
        same = same && std::memcmp(k, p, n * sizeof(double)) == 0;
        k += n;
    });
    return same;
}
// the smallest tested k with |M^k|_inf <= 2^-60 (squarings, then the lower bits of the exponent); -1: not within 2^20 steps
long long halo_of(int d, const std::vector<double>& M) {
    const size_t dd = (size_t)d * d;
    const double thr = std::ldexp(1.0, -60);
    std::vector<std::vector<double>> pw;
    pw.push_back(M);
    int j = 0;
    while (norm_inf(d, pw.back().data()) > thr) {
        if (j >= 20) return -1;
        std::vector<double> sq(dd);
        matmul(d, pw.back().data(), pw.back().data(), sq.data());
        pw.push_back(std::move(sq));
        ++j;
    }
    long long halo = 1LL << j;
    if (j >= 2) {
        std::vector<double> cur = pw[j - 1], cand(dd);
        long long ex = 1LL << (j - 1);
        const int bmin = std::max(0, j - 5);
        for (int b = j - 2; b >= bmin; --b) {
            matmul(d, cur.data(), pw[b].data(), cand.data());
            if (norm_inf(d, cand.data()) > thr) {
                cur = cand;
                ex += 1LL << b;
            }
        }
        halo = ex + (1LL << bmin);
    }
    return halo;
}
}  // namespace

bool plan(Engine* e, const ModelHost& m, long long T) {
    if (same_model(e, m, T)) {
        e->info.plan_ms = 0.0;
        return e->info.why == kOk;
    }
    static const bool cpu_ok = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");      // (this object's host code is built with both)
    if (!cpu_ok) {
        e->info = Info{};
        e->info.why = kAlloc;
        return false;
    }
    const auto t_begin = std::chrono::steady_clock::now();
    e->have = false;
    e->dev_current = false;
    e->post_ready = false;
    e->info = Info{};
    const int d = m.d;
    const size_t dd = (size_t)d * d;
    e->d = d;
    e->dp = d <= 31 ? 32 : 64;
    e->kname = e->dp == 32 ? (dpp_enabled() ? (d <= 15 ? "k_wide_lml4<16>" : "k_wide_lml4") : "k_wide_lml<32>") : (d <= 47 && dpp_enabled() ? "k_wide_lml4<48>" : "k_wide_lml<64>");
    e->A.assign(dd, 0.0);
    std::vector<double> Q(dd), P(dd), AP(dd), Pp(dd), Pf(dd), v(d);
    for (int i = 0; i < d; ++i)
        for (int k = 0; k < d; ++k) {
            e->A[(size_t)i * d + k] = m.A[i + (size_t)k * d];
            Q[(size_t)i * d + k] = 0.5 * (m.Q[i + (size_t)k * d] + m.Q[k + (size_t)i * d]);
            P[(size_t)i * d + k] = 0.5 * (m.x0P[i + (size_t)k * d] + m.x0P[k + (size_t)i * d]);
        }
    e->avec.assign(m.a, m.a + d);
    e->hvec.assign(m.H, m.H + d);
    e->x0m.assign(m.x0m, m.x0m + d);
    e->hh = m.hh;
    e->R = m.R;
    const double* A = e->A.data();
    const double* h = e->hvec.data();
    auto done = [&](int why) {
        e->info.why = why;
        e->info.plan_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
        // (a plan that declined is kept as well: the next call of the same model asks no more than the memcmp)
        e->key.clear();
        model_words(m, [&](const double* p, size_t n) { e->key.insert(e->key.end(), p, p + n); });
        e->key_T = T;
        e->have = true;
        return why == kOk;
    };
    // ---- the covariance half of lgssm.jl:99-165 to its fixed point: P <- A P A' + Q; S = h' P h + R; K = P h / S; P <- P - K S K'
    e->Kt.clear();
    e->St.clear();
    e->Pf_head.clear();
    bool keep_pf = true;
    e->sum_logS_head = 0.0;
    int n0 = -1;
    double prev_chg = 1e300, rate = 0.0, prev_S = 0.0;
    for (int t = 0; t < kHeadMax; ++t) {
        matmul(d, A, P.data(), AP.data());
        double scale = 0.0;
        for (int i = 0; i < d; ++i)
            for (int j = 0; j <= i; ++j) {
                const double *x = AP.data() + (size_t)i * d, *yv = A + (size_t)j * d;
                double s = Q[(size_t)i * d + j];
                for (int k = 0; k < d; ++k) s += x[k] * yv[k];
                Pp[(size_t)i * d + j] = Pp[(size_t)j * d + i] = s;
            }
        double S = m.R;
        for (int i = 0; i < d; ++i) {
            double s = 0.0;
            for (int k = 0; k < d; ++k) s += Pp[(size_t)i * d + k] * h[k];
            v[i] = s;
            S += h[i] * s;
        }
        if (!(S > 0.0) || !std::isfinite(S)) return done(kNotPD);
        const double iS = 1.0 / S;
        double chg = 0.0;
        for (int i = 0; i < d; ++i)
            for (int j = 0; j <= i; ++j) {
                const double pf = Pp[(size_t)i * d + j] - v[i] * v[j] * iS;
                chg = std::max(chg, std::fabs(pf - P[(size_t)i * d + j]));
                scale = std::max(scale, std::fabs(pf));
                Pf[(size_t)i * d + j] = Pf[(size_t)j * d + i] = pf;
            }
        for (int i = 0; i < d; ++i) e->Kt.push_back(v[i] * iS);
        e->St.push_back(S);
        e->sum_logS_head += std::log(S);
        if (keep_pf) {
            if (e->Pf_head.size() + dd > ((size_t)64 << 20) / sizeof(double)) {      // (64 MB of head covariances at most: beyond, _filter is the dense engine's)
                keep_pf = false;
                e->Pf_head.clear();
            } else {
                e->Pf_head.insert(e->Pf_head.end(), Pf.begin(), Pf.end());
            }
        }
        P.swap(Pf);
        // the contraction rate, while the changes are still well above rounding: "no longer moves" bounds the DISTANCE to the fixed point by
        // (last change) / (1 - rate) only -- a recursion that creeps (fine spacings: rate -> 1) is not settled when its steps fall to the rounding floor
        if (chg > 1e-11 * scale && prev_chg > 1e-11 * scale && chg < prev_chg) rate = chg / prev_chg;
        // settled: the step changes nothing at all, or nothing beyond rounding -- a few ulps of the largest entry, or no longer shrinking at the rounding
        // floor -- with the fixed point within 1e-12 of it by that bound
        // (... and the innovation variance -- what the log-likelihood sees -- within 1e-13 of ITS fixed point: S can be orders of magnitude below the
        //  covariance's largest entry)
        const bool still = chg <= 4.0 * 2.220446049250313e-16 * scale || (t >= 16 && chg >= prev_chg && chg <= 1e-13 * scale);
        const double dS = std::fabs(S - prev_S);
        if (chg == 0.0 || (still && chg <= (1.0 - rate) * 1e-12 * scale && dS <= (1.0 - rate) * 1e-13 * S)) {
            n0 = t + 1;
            break;
        }
        prev_chg = chg;
        prev_S = S;
    }
    if (n0 < 0) return done(kNotSettled);
    e->info.n0 = n0;
    e->info.nhs = n0;
    e->Sss = e->St.back();
    const double* K = e->Kt.data() + (size_t)(n0 - 1) * d;
    // ---- Phi = (I - K h') A = A - K g', g = A' h; c = a - K g0, g0 = h . a
    std::vector<double> g(d, 0.0), Phi(dd);
    for (int k = 0; k < d; ++k)
        for (int j = 0; j < d; ++j) g[j] += h[k] * A[(size_t)k * d + j];
    double g0 = 0.0;
    for (int k = 0; k < d; ++k) g0 += h[k] * e->avec[k];
    e->g0 = g0;
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) Phi[(size_t)i * d + j] = A[(size_t)i * d + j] - K[i] * g[j];
    const long long halo = halo_of(d, Phi);
    if (halo < 0) return done(kSlowMixing);
    e->info.halo = (int)halo;
    const long long Tb = T - n0;      // steps behind the head
    if (Tb < 64) return done(kTooShort);
    // chunks: as many waves as the chip holds several times over, none shorter than 64 steps
    // chunks: one wave's worth of them per SIMD (4096) at least -- none shorter than 64 steps --, and more (up to 16384: the stalls of a wave's dependent
    // DPP multiply-adds are another wave's issue slots) once a chunk is still four halos long: 12 % at T = 1e7 (scripts/r06_mid_d_time.py)
    static const long long forced_chunks = [] {      // TGP_WIDE_CHUNKS=<n>: development
        const char* sv = std::getenv("TGP_WIDE_CHUNKS");
        const long long v = sv ? std::atoll(sv) : 0;
        return v >= 1 && v <= 65536 ? v : 0ll;
    }();
    long long want = std::max<long long>(kMaxChunks, std::min<long long>(4 * (long long)kMaxChunks, Tb / (4 * std::max<long long>(halo, 16))));
    if (forced_chunks) want = forced_chunks;
    long long chunks = std::min<long long>(want, Tb / 64);
    // a slowly mixing closed loop: no chunk shorter than half its warm-up (else the warm-ups are most of the work; a chunk within `halo` of the head starts
    // from the head's own end state whatever its length)
    chunks = std::max<long long>(1, std::min<long long>(chunks, Tb / std::max<long long>(64, halo / 2)));
    long long len = (Tb + chunks - 1) / chunks;
    chunks = (Tb + len - 1) / len;
    e->info.chunks = chunks;
    e->info.chunk_len = len;
    // ---- the forward kernel's table
    const int DP = e->dp;
    e->tab_host.assign((size_t)(DP + 2) * 64, 0.0);
    for (int i = 0; i < d; ++i) {
        for (int j = 0; j < d; ++j) e->tab_host[(size_t)j * 64 + i] = Phi[(size_t)i * d + j];
        e->tab_host[(size_t)DP * 64 + i] = K[i];
        e->tab_host[(size_t)(DP + 1) * 64 + i] = e->avec[i] - K[i] * g0;
    }
    for (int j = 0; j < d; ++j) e->tab_host[(size_t)j * 64 + d] = -g[j];      // the observer: r = u - g . z - g0
    e->tab_host[(size_t)DP * 64 + d] = 1.0;
    e->tab_host[(size_t)(DP + 1) * 64 + d] = -g0;
    return done(kOk);
}

namespace {
// The posterior half of the plan, data-free as the rest: Psi = (I - h K') A' and gw = R A K of the stationary step, halo_back, the partial sums
// q_j = sum_(k < j) (gw' Psi^k h)^2 / S of the variance's quadratic form at the series' end (n1 of them until they no longer change), Lam_inf = the
// fixed point of Lam = h h' / S + Psi Lam Psi' by doubling, and from it the head's variances backwards through the head's own steps.
int plan_post(Engine* e, long long T) {
    const int d = e->d, DP = e->dp, n0 = e->info.n0;
    const size_t dd = (size_t)d * d;
    const double *A = e->A.data(), *h = e->hvec.data();
    const double R = e->R, S = e->Sss;
    const double* K = e->Kt.data() + (size_t)(n0 - 1) * d;
    auto AK_of = [&](const double* Kv, std::vector<double>& out) {
        for (int j = 0; j < d; ++j) {
            double s = 0.0;
            for (int k = 0; k < d; ++k) s += A[(size_t)j * d + k] * Kv[k];
            out[j] = s;
        }
    };
    auto psi_of = [&](const std::vector<double>& AK, std::vector<double>& Psi) {      // Psi[i][j] = A[j][i] - h_i (A K)_j
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j) Psi[(size_t)i * d + j] = A[(size_t)j * d + i] - h[i] * AK[j];
    };
    std::vector<double> AK(d), Psi(dd), gw(d);
    AK_of(K, AK);
    psi_of(AK, Psi);
    for (int j = 0; j < d; ++j) gw[j] = R * AK[j];
    const long long hb = halo_of(d, Psi);
    if (hb < 0) return kSlowMixing;
    e->info.halo_back = (int)hb;
    // ---- the series' end: q_j
    e->qtab.assign(1, 0.0);
    {
        std::vector<double> u(h, h + d), un(d);
        double gw1 = 0.0;
        for (int j = 0; j < d; ++j) gw1 += std::fabs(gw[j]);
        double q = 0.0;
        long long n1 = -1;
        for (long long k = 0; k < kTailMax; ++k) {
            double c = 0.0, umax = 0.0;
            for (int j = 0; j < d; ++j) {
                c += gw[j] * u[j];
                umax = std::max(umax, std::fabs(u[j]));
            }
            q += c * c / S;
            e->qtab.push_back(q);
            const double bound = gw1 * umax;      // |gw' Psi^k' h| for every later k' is below this times |Psi^(k' - k)|
            if (bound * bound / S <= 1e-20 * std::max(q, 1e-300) && k >= 2) {
                n1 = k + 1;
                break;
            }
            for (int i = 0; i < d; ++i) {
                double s2 = 0.0;
                for (int j = 0; j < d; ++j) s2 += Psi[(size_t)i * d + j] * u[j];
                un[i] = s2;
            }
            u.swap(un);
        }
        if (n1 < 0) return kTailLong;
        e->info.n1 = (int)n1;
        e->qinf = q;
        if ((long long)n0 + n1 + 1 > T) return kTooShort;
    }
    e->vbase = (S - R) * R / S;
    // ---- Lam_inf by doubling: Lam <- Lam + M Lam M', M <- M^2
    std::vector<double> Lam(dd), M(Psi), T1(dd), T2(dd);
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) Lam[(size_t)i * d + j] = h[i] * h[j] / S;
    for (int it = 0; it < 40; ++it) {
        if (norm_inf(d, M.data()) <= 1e-12) break;
        matmul(d, M.data(), Lam.data(), T1.data());      // T1 = M Lam
        for (int i = 0; i < d; ++i)
            for (int j = 0; j <= i; ++j) {               // Lam += T1 M'
                double s2 = 0.0;
                for (int k = 0; k < d; ++k) s2 += T1[(size_t)i * d + k] * M[(size_t)j * d + k];
                T2[(size_t)i * d + j] = T2[(size_t)j * d + i] = s2;
            }
        for (size_t i = 0; i < dd; ++i) Lam[i] += T2[i];
        matmul(d, M.data(), M.data(), T1.data());
        M.swap(T1);
    }
    // ---- the head's variances, backwards through its own steps (row n0 - 1 first: Lam_(n0) = Lam_inf)
    e->headvar.assign(n0, 0.0);
    {
        std::vector<double> AKt(d), gwt(d), Pt(dd);
        for (int t = n0 - 1; t >= 0; --t) {
            const double* Kv = e->Kt.data() + (size_t)t * d;
            const double St = e->St[t];
            AK_of(Kv, AKt);
            for (int j = 0; j < d; ++j) gwt[j] = R * AKt[j];
            double qf = 0.0;
            for (int i = 0; i < d; ++i) {
                double s2 = 0.0;
                for (int j = 0; j < d; ++j) s2 += Lam[(size_t)i * d + j] * gwt[j];
                qf += gwt[i] * s2;
            }
            e->headvar[t] = (St - R) * R / St - qf;
            if (t == 0) break;
            psi_of(AKt, Pt);
            matmul(d, Pt.data(), Lam.data(), T1.data());
            for (int i = 0; i < d; ++i)
                for (int j = 0; j <= i; ++j) {
                    double s2 = h[i] * h[j] / St;
                    for (int k = 0; k < d; ++k) s2 += T1[(size_t)i * d + k] * Pt[(size_t)j * d + k];
                    T2[(size_t)i * d + j] = T2[(size_t)j * d + i] = s2;
                }
            Lam.swap(T2);
        }
    }
    // ---- the backward kernel's table
    e->tabb_host.assign((size_t)(DP + 1) * 64, 0.0);
    for (int i = 0; i < d; ++i) {
        for (int j = 0; j < d; ++j) e->tabb_host[(size_t)j * 64 + i] = Psi[(size_t)i * d + j];
        e->tabb_host[(size_t)DP * 64 + i] = h[i] / S;
    }
    for (int j = 0; j < d; ++j) e->tabb_host[(size_t)j * 64 + d] = gw[j];      // the observer: mean_t - y_t = gw . lam_(t+1) - (R / S) r_t
    e->tabb_host[(size_t)DP * 64 + d] = -R / S;
    return kOk;
}
}  // namespace

bool plan_posterior(Engine* e, long long T) {
    if (!e->have || e->info.why != kOk) return false;
    if (!e->post_ready) {
        const auto t_begin = std::chrono::steady_clock::now();
        e->post_why = plan_post(e, T);
        e->post_ready = true;
        e->dev_current = false;
        e->info.plan_post_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    } else {
        e->info.plan_post_ms = 0.0;
    }
    e->info.why_post = e->post_why;
    return e->post_why == kOk;
}

int run(Engine* e, hipStream_t stream, const Call& c, double* lml_out, std::string* err) {
    auto fail = [&](hipError_t rc, const char* what) {
        if (err) *err = std::string("tgp_wide: ") + what + ": " + hipGetErrorString(rc);
        return (int)rc;
    };
    const bool post = c.mean != nullptr;
    if (!e->have || e->info.why != kOk || (post && (!e->post_ready || e->post_why != kOk || !c.var || !c.Rnew))) return fail(hipErrorInvalidValue, "no plan");
    const int d = e->d, DP = e->dp, n0 = e->info.n0;
    const long long T = c.T;
    hipError_t rc;
    if (!e->pinned) {
        rc = tgp_alloc::host_malloc(reinterpret_cast<void**>(&e->pinned), (size_t)(5 * kHeadMax + 64 + 65536) * sizeof(double), hipHostMallocDefault);
        if (rc != hipSuccess) return fail(rc, "pinned buffer");
    }
    // device tables: forward | backward | qtab
    const size_t nf = e->tab_host.size(), nb = post ? e->tabb_host.size() : 0, nq = post ? e->qtab.size() : 0;
    const size_t need = (nf + (size_t)(DP + 1) * 64 + (size_t)kTailMax + 2) * sizeof(double);
    if (need > e->dev_cap) {
        if (e->dev) (void)tgp_alloc::dev_free(e->dev);
        e->dev = nullptr;
        e->dev_cap = 0;
        rc = tgp_alloc::dev_malloc(reinterpret_cast<void**>(&e->dev), need);
        if (rc != hipSuccess) return fail(rc, "tables");
        e->dev_cap = need;
        e->dev_current = false;
    }
    double *tab_f = e->dev, *tab_b = e->dev + nf, *qtab_d = tab_b + (size_t)(DP + 1) * 64;
    if (!e->dev_current || (post && !e->dev_post_current)) {
        rc = hipMemcpyAsync(tab_f, e->tab_host.data(), nf * sizeof(double), hipMemcpyHostToDevice, stream);
        if (rc == hipSuccess && post) rc = hipMemcpyAsync(tab_b, e->tabb_host.data(), nb * sizeof(double), hipMemcpyHostToDevice, stream);
        if (rc == hipSuccess && post) rc = hipMemcpyAsync(qtab_d, e->qtab.data(), nq * sizeof(double), hipMemcpyHostToDevice, stream);
        if (rc != hipSuccess) return fail(rc, "table upload");
        e->dev_current = true;
        e->dev_post_current = post;
    }
    if (post && (size_t)T * sizeof(double) > e->rbuf_cap) {
        if (e->rbuf) (void)tgp_alloc::dev_free(e->rbuf);
        e->rbuf = nullptr;
        e->rbuf_cap = 0;
        rc = tgp_alloc::dev_malloc(reinterpret_cast<void**>(&e->rbuf), (size_t)T * sizeof(double));
        if (rc != hipSuccess) return fail(rc, "innovation buffer");
        e->rbuf_cap = (size_t)T * sizeof(double);
    }
    double *yh = e->pinned, *Rh = yh + kHeadMax, *mh = Rh + kHeadMax, *vh = mh + kHeadMax, *hth = vh + kHeadMax, *lam = hth + kHeadMax, *part = lam + 64;
    rc = hipMemcpyAsync(yh, c.y, (size_t)n0 * sizeof(double), hipMemcpyDeviceToHost, stream);
    if (rc == hipSuccess && post) rc = hipMemcpyAsync(Rh, c.Rnew, (size_t)(c.rnew_per_step ? n0 : 1) * sizeof(double), hipMemcpyDeviceToHost, stream);
    if (rc == hipSuccess && c.h_t) rc = hipMemcpyAsync(hth, c.h_t, (size_t)n0 * sizeof(double), hipMemcpyDeviceToHost, stream);
    if (rc != hipSuccess) return fail(rc, "head observations");
    rc = hipStreamSynchronize(stream);
    if (rc != hipSuccess) return fail(rc, "head observations");
    // ---- the head forward: lgssm.jl:147-165 with the plan's gains
    ZArg z0;
    for (int i = 0; i < 64; ++i) z0.z[i] = 0.0;
    double quad = 0.0;
    const double *A = e->A.data(), *h = e->hvec.data();
    e->head_r.resize(n0);
    e->head_m.clear();
    {
        std::vector<double> mcur(e->x0m), mp(d);
        for (int t = 0; t < n0; ++t) {
            double pred = c.h_t ? hth[t] : e->hh;
            for (int i = 0; i < d; ++i) {
                double s = e->avec[i];
                const double* ai = A + (size_t)i * d;
                for (int k = 0; k < d; ++k) s += ai[k] * mcur[k];
                mp[i] = s;
                pred += h[i] * s;
            }
            const double r = yh[t] - pred;
            e->head_r[t] = r;
            quad += r * r / e->St[t];
            const double* K = e->Kt.data() + (size_t)t * d;
            for (int i = 0; i < d; ++i) mcur[i] = mp[i] + K[i] * r;
            if (c.fm) e->head_m.insert(e->head_m.end(), mcur.begin(), mcur.end());
        }
        for (int i = 0; i < d; ++i) z0.z[i] = mcur[i];
    }
    if (c.fm) {      // _filter: the head's means and covariances, the settled covariance behind them
        if (!c.fP || e->Pf_head.size() != (size_t)n0 * d * d) return fail(hipErrorInvalidValue, "filter outputs");
        rc = hipMemcpyAsync(c.fm, e->head_m.data(), (size_t)n0 * d * sizeof(double), hipMemcpyHostToDevice, stream);
        if (rc == hipSuccess) rc = hipMemcpyAsync(c.fP, e->Pf_head.data(), e->Pf_head.size() * sizeof(double), hipMemcpyHostToDevice, stream);
        if (rc != hipSuccess) return fail(rc, "head filter outputs");
        const long long nfill = (T - n0) * (long long)d * d;
        const unsigned blocks = (unsigned)std::min<long long>((nfill + 255) / 256, 16384);
        if (nfill > 0) hipLaunchKernelGGL(k_wide_fill_cov, dim3(blocks), dim3(256), 0, stream, c.fP, (long long)n0, T, d * d);
    }
    const long long chunks = e->info.chunks;
    double* rout = post ? e->rbuf : nullptr;
    const bool four = DP == 32 && dpp_enabled();
    const unsigned grid4 = (unsigned)((chunks + 3) / 4);
    const bool three = DP == 64 && d <= 47 && dpp_enabled();      // three components per lane
    const bool one = d <= 15;      // one component per lane
#define TGP_WIDE_LML4(KEEP, NB) \
    hipLaunchKernelGGL((k_wide_lml4<KEEP, NB>), dim3(grid4), dim3(64), 0, stream, tab_f, c.y, e->hh, T, (long long)n0, e->info.chunk_len, (long long)e->info.halo, chunks, d, z0, part, rout, c.h_t, c.fm)
    if (three && post)
        hipLaunchKernelGGL(k_wide_lml43<true>, dim3(grid4), dim3(64), 0, stream, tab_f, c.y, e->hh, T, (long long)n0, e->info.chunk_len, (long long)e->info.halo, chunks, d, z0, part, rout,
                           c.h_t, c.fm);
    else if (three)
        hipLaunchKernelGGL(k_wide_lml43<false>, dim3(grid4), dim3(64), 0, stream, tab_f, c.y, e->hh, T, (long long)n0, e->info.chunk_len, (long long)e->info.halo, chunks, d, z0, part, rout,
                           c.h_t, c.fm);
    else if (four && post && one) TGP_WIDE_LML4(true, 1);
    else if (four && post) TGP_WIDE_LML4(true, 2);
    else if (four && one) TGP_WIDE_LML4(false, 1);
    else if (four) TGP_WIDE_LML4(false, 2);
#undef TGP_WIDE_LML4
    else if (DP == 32)
        hipLaunchKernelGGL(k_wide_lml<32>, dim3((unsigned)chunks), dim3(64), 0, stream, tab_f, c.y, e->hh, T, (long long)n0, e->info.chunk_len, (long long)e->info.halo, d, z0, part,
                           rout, c.h_t, c.fm, d);
    else
        hipLaunchKernelGGL(k_wide_lml<64>, dim3((unsigned)chunks), dim3(64), 0, stream, tab_f, c.y, e->hh, T, (long long)n0, e->info.chunk_len, (long long)e->info.halo, d, z0, part,
                           rout, c.h_t, c.fm, d);
    rc = hipGetLastError();
    if (rc != hipSuccess) return fail(rc, "launch");
    if (post) {
        if (three)
            hipLaunchKernelGGL(k_wide_bwd43, dim3(grid4), dim3(64), 0, stream, tab_b, c.y, e->rbuf, c.Rnew, c.rnew_per_step, qtab_d, (long long)e->info.n1, e->vbase, e->qinf, T,
                               (long long)n0, e->info.chunk_len, (long long)e->info.halo_back, chunks, d, c.mean, c.var, lam);
        else if (four && one)
            hipLaunchKernelGGL(k_wide_bwd4<1>, dim3(grid4), dim3(64), 0, stream, tab_b, c.y, e->rbuf, c.Rnew, c.rnew_per_step, qtab_d, (long long)e->info.n1, e->vbase, e->qinf, T,
                               (long long)n0, e->info.chunk_len, (long long)e->info.halo_back, chunks, d, c.mean, c.var, lam);
        else if (four)
            hipLaunchKernelGGL(k_wide_bwd4<2>, dim3(grid4), dim3(64), 0, stream, tab_b, c.y, e->rbuf, c.Rnew, c.rnew_per_step, qtab_d, (long long)e->info.n1, e->vbase, e->qinf, T,
                               (long long)n0, e->info.chunk_len, (long long)e->info.halo_back, chunks, d, c.mean, c.var, lam);
        else if (DP == 32)
            hipLaunchKernelGGL(k_wide_bwd<32>, dim3((unsigned)chunks), dim3(64), 0, stream, tab_b, c.y, e->rbuf, c.Rnew, c.rnew_per_step, qtab_d, (long long)e->info.n1, e->vbase,
                               e->qinf, T, (long long)n0, e->info.chunk_len, (long long)e->info.halo_back, d, d, c.mean, c.var, lam);
        else
            hipLaunchKernelGGL(k_wide_bwd<64>, dim3((unsigned)chunks), dim3(64), 0, stream, tab_b, c.y, e->rbuf, c.Rnew, c.rnew_per_step, qtab_d, (long long)e->info.n1, e->vbase,
                               e->qinf, T, (long long)n0, e->info.chunk_len, (long long)e->info.halo_back, d, d, c.mean, c.var, lam);
        rc = hipGetLastError();
        if (rc != hipSuccess) return fail(rc, "launch");
    }
    rc = hipStreamSynchronize(stream);
    if (rc != hipSuccess) return fail(rc, "kernel");
    if (post) {
        // ---- the head backward: lam_t = h r_t / S_t + Psi_t lam_(t+1), Psi_t lam = A' lam - h (A K_t) . lam; mean_t = y_t - (R / S_t) r_t + R (A K_t) . lam_(t+1)
        std::vector<double> lcur(lam, lam + d), AKt(d), ln(d);
        for (int t = n0 - 1; t >= 0; --t) {
            const double* Kv = e->Kt.data() + (size_t)t * d;
            const double St = e->St[t], r = e->head_r[t];
            double akl = 0.0;
            for (int j = 0; j < d; ++j) {
                double s = 0.0;
                for (int k = 0; k < d; ++k) s += A[(size_t)j * d + k] * Kv[k];
                AKt[j] = s;
                akl += s * lcur[j];
            }
            mh[t] = yh[t] - (e->R / St) * r + e->R * akl;
            vh[t] = e->headvar[t] + Rh[c.rnew_per_step ? t : 0];
            for (int i = 0; i < d; ++i) {
                double s = h[i] * (r / St - akl);
                for (int k = 0; k < d; ++k) s += A[(size_t)k * d + i] * lcur[k];
                ln[i] = s;
            }
            lcur.swap(ln);
        }
        rc = hipMemcpyAsync(c.mean, mh, (size_t)n0 * sizeof(double), hipMemcpyHostToDevice, stream);
        if (rc == hipSuccess) rc = hipMemcpyAsync(c.var, vh, (size_t)n0 * sizeof(double), hipMemcpyHostToDevice, stream);
        if (rc == hipSuccess) rc = hipStreamSynchronize(stream);
        if (rc != hipSuccess) return fail(rc, "head outputs");
    }
    double ssq = 0.0;
    for (long long k = 0; k < chunks; ++k) ssq += part[k];
    const double kLog2Pi = 1.8378770664093454835606594728112;
    *lml_out = -0.5 * ((double)T * kLog2Pi + e->sum_logS_head + (double)(T - n0) * std::log(e->Sss) + quad + ssq / e->Sss);
    return 0;
}

int rand(Engine* e, hipStream_t stream, const ModelHost& m, long long T, const double* x0_host, const double* eps_t, const double* eps_e, double* y_out, bool* declined,
         std::string* err) {
    *declined = true;
    auto fail = [&](hipError_t rc, const char* what) {
        if (err) *err = std::string("tgp_wide: ") + what + ": " + hipGetErrorString(rc);
        return (int)rc;
    };
    static const bool cpu_ok = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
    const int d = m.d;
    if (!cpu_ok || !supports(d) || T < 256) return 0;
    const size_t dd = (size_t)d * d;
    const int DP = d <= 31 ? 32 : 64;
    // row-major A; U = chol(Q + 1e-9 I) (upper, row-major); the open loop's halo
    std::vector<double> A(dd), U(dd, 0.0);
    for (int i = 0; i < d; ++i)
        for (int k = 0; k < d; ++k) A[(size_t)i * d + k] = m.A[i + (size_t)k * d];
    for (int j = 0; j < d; ++j)
        for (int i = 0; i <= j; ++i) {
            double acc = 0.5 * (m.Q[i + (size_t)j * d] + m.Q[j + (size_t)i * d]) + (i == j ? 1e-9 : 0.0);
            for (int k = 0; k < i; ++k) acc -= U[(size_t)k * d + i] * U[(size_t)k * d + j];
            if (i == j) {
                if (!(acc > 0.0)) return 0;      // (not positive definite: the engines of before report it)
                U[(size_t)j * d + j] = std::sqrt(acc);
            } else {
                U[(size_t)i * d + j] = acc / U[(size_t)i * d + i];
            }
        }
    const long long halo = halo_of(d, A);
    if (halo < 0 || halo > T / 4) return 0;      // (an open loop that does not forget -- ApproxPeriodicKernel() alone: |lambda| = 1 -- or hardly)
    long long chunks = std::min<long long>(kMaxChunks, std::max<long long>(1, T / std::max<long long>(64, halo / 2)));
    const long long len = (T + chunks - 1) / chunks;
    chunks = (T + len - 1) / len;
    std::vector<double> tab((size_t)(2 * DP + 1) * 64, 0.0);
    double ha = m.hh;
    for (int i = 0; i < d; ++i) ha += m.H[i] * m.a[i];
    for (int i = 0; i < d; ++i) {
        for (int j = 0; j < d; ++j) {
            tab[(size_t)j * 64 + i] = A[(size_t)i * d + j];
            tab[(size_t)(DP + j) * 64 + i] = U[(size_t)j * d + i];      // U'[i][j]
        }
        tab[(size_t)(2 * DP) * 64 + i] = m.a[i];
    }
    for (int j = 0; j < d; ++j) {
        double gj = 0.0, uh = 0.0;
        for (int i = 0; i < d; ++i) {
            gj += m.H[i] * A[(size_t)i * d + j];
            uh += U[(size_t)j * d + i] * m.H[i];
        }
        tab[(size_t)j * 64 + d] = gj;
        tab[(size_t)(DP + j) * 64 + d] = uh;
    }
    tab[(size_t)(2 * DP) * 64 + d] = ha;
    hipError_t rc;
    const size_t need = tab.size() * sizeof(double);
    if (need > e->rand_cap) {
        if (e->rand_dev) (void)tgp_alloc::dev_free(e->rand_dev);
        e->rand_dev = nullptr;
        e->rand_cap = 0;
        rc = tgp_alloc::dev_malloc(reinterpret_cast<void**>(&e->rand_dev), need);
        if (rc != hipSuccess) return fail(rc, "rand table");
        e->rand_cap = need;
    }
    rc = hipMemcpyAsync(e->rand_dev, tab.data(), need, hipMemcpyHostToDevice, stream);
    if (rc == hipSuccess) rc = hipStreamSynchronize(stream);      // (tab is a temporary)
    if (rc != hipSuccess) return fail(rc, "rand table upload");
    ZArg x0;
    for (int i = 0; i < 64; ++i) x0.z[i] = i < d ? x0_host[i] : 0.0;
    const double sqrtR = std::sqrt(m.R);
    if (DP == 32)
        hipLaunchKernelGGL(k_wide_rand<32>, dim3((unsigned)chunks), dim3(64), 0, stream, e->rand_dev, eps_t, eps_e, sqrtR, T, len, halo, d, d, x0, y_out);
    else
        hipLaunchKernelGGL(k_wide_rand<64>, dim3((unsigned)chunks), dim3(64), 0, stream, e->rand_dev, eps_t, eps_e, sqrtR, T, len, halo, d, d, x0, y_out);
    rc = hipGetLastError();
    if (rc != hipSuccess) return fail(rc, "launch");
    *declined = false;
    return 0;
}

}  // namespace tgp_wide
