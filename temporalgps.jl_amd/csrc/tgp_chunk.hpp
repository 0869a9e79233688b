// Per-lane work of the time-parallel Kalman engine: every lane owns one CHUNK of L0 consecutive
// processing steps and runs it sequentially with all matrices in registers.
//
//   pass 1  chunk_reduce_*   steps of a chunk  -> one scan element        (cheap rank-one extension)
//   (block scans over the chunk elements: tgp_kernels.hpp)
//   pass 2  chunk_apply_*    chunk carry-in state -> the reference's own sequential recursion
//                            (predict / posterior_and_lml / invert_dynamics, bit-for-bit the same
//                            arithmetic as the CPU path once the carry-in is known)
//   pass 3  chunk_smooth     RTS smoother = Reverse step_marginals on the posterior model
//
// "Processing index" r runs 0..T-1 in the order the reference visits steps:
//   Forward (gauss_markov_model.jl:38):  r -> storage index r,      step = predict(r) then emission(r)
//   Reverse (gauss_markov_model.jl:40):  r -> storage index T-1-r,  step = emission(T-1-r) then predict(T-1-r)
// A Reverse model is run as "predict(T-r) [skipped at r = 0] then emission(T-1-r)", which is the same
// sequence of operations (lgssm.jl:111-115, 161-165, 183-187); the trailing predict only feeds the
// returned final state, which no caller on this path reads.
#pragma once
#include "tgp_math.hpp"

namespace TGP_NS {

struct ModelView {
    int64_t T;         // number of PROCESSING steps = Tt * p (one scalar observation each)
    int32_t ordering;  // 0 = Forward, 1 = Reverse
    int32_t p;         // observations per time step (1: ScalarOutputLGC; > 1: SmallOutputLGC with diagonal R, run as p
                       // consecutive scalar updates -- algebraically the joint update of lgc.jl:129-141)
    const double* A;   // [Tt|1][d*d] column-major
    const double* a;   // [Tt|1][d]
    const double* Q;   // [Tt|1][d*d]
    const double* H;   // [Tt|1][p][d]   row j = j-th observation functional (emission.A[j, :])
    const double* h;   // [Tt|1][p]
    const double* R;   // [Tt|1][p]      diagonal of the emission noise
    int64_t sA, sa, sQ, sH, sh, sR;  // stride in doubles per TIME step; 0 == Fill (shared)
    const double* y;                 // [Tt][p]
    const uint8_t* missing;          // [Tt][p] or nullptr; 1 => y := 0, R := 1e15 (missings.jl:55-101, lgc.jl:143-151)
    // Time-tiled copies of the PER-STEP model arrays (general layout), in processing order, one lane per chunk:
    //   tile_t[ ((chunk/64 * Lt + time_in_chunk) * nc_t + component) * 64 + chunk%64 ]   transitions (A, a, Q)
    //   tile_e[ ((chunk/64 * L0 + step_in_chunk) * nc_e + component) * 64 + chunk%64 ]   emissions (H row, h, R)
    // so that a wave (64 consecutive chunks) reads one component of one step as ONE coalesced 512-byte row.
    // tile_mask says which arrays are tiled; the others are shared (Fill) and read from the pointers above.
    const double* tile_t;
    const double* tile_e;
    int32_t nc_t, nc_e;
    uint32_t tile_mask;
    int32_t small_out;   // emissions are SmallOutputLGC (affects rand: lgc.jl:84-87 adds 1e-9 to the noise)
    int64_t Tt;          // time steps
    // Tangents of the SHARED model blocks w.r.t. ONE hyper-parameter (forward-mode gradient pass, namespace
    // tgp::ad); null => zero. Same shapes as A, a, Q, H, h, R above (one block each).
    const double *dA, *da, *dQ, *dH, *dh, *dR;
    // Tangent of the tiled transition record (general layout, gradient pass): same layout as tile_t; null => zero.
    const double* tile_t_tan;
    // Stationary-covariance record of the posterior path (see "stationary covariance" in tgp_chunk_body.inc), [1 + 2 DS][n0]:
    // row 0 = first step of the chunk that pass 2 ran in the mean-only form (as a double; >= L0: none), rows 1.. = the packed
    // upper triangles of the two filtering covariances the chunk alternates between from there on. Non-null enables the
    // mean-only steps of passes 2 and 3 (the API sets it for shared-layout models with one shared R, p = 1, no missing data, d <= 4).
    double* steady;
    // Closed-form SDE transitions (tile_mask & kTileDt; the API sets it for models given by tgp_model_set_sde whose drift matrix is
    // block diagonal with ONE eigenvalue -lambda_b per block and nilpotency <= 3 -- sums of scaled / stretched Matern-1/2, -3/2, -5/2
    // terms): the transition record holds ONE double per step, tau_k = t_k - t_{k-1} (< 0: the first transition), and the loader
    // evaluates A_k = exp(F tau_k) = e^{-lambda tau} (I + tau N + tau^2 / 2 N^2) per block and Q_k = Pinf - A_k Pinf A_k' in
    // registers (lti_sde.jl:135-146) -- the passes stream 8 B per step instead of 16 d^2.
    //   sde = [ lambda per row: d | N: d^2 | N^2 / 2: d^2 | Pinf: d^2 | A1: d^2 | Q1: d^2 ]   (column-major, wave-uniform loads)
    const double* sde;
    int32_t sde_first;   // 1: the first transition is (A1, Q1) as given; 0: the reference's dt_1 := 1
};

// Wave-level helpers of the stationary-covariance steps: a vote over the ACTIVE lanes of the wave, and a wave-uniform integer
// the compiler may keep in an SGPR. The host emulation runs one chunk at a time.
#if defined(__HIP_DEVICE_COMPILE__)
#define TGP_WAVE_ALL(pred) (__builtin_amdgcn_ballot_w64(!(pred)) == 0ull)
#define TGP_WAVE_UNIFORM_INT(x) __builtin_amdgcn_readfirstlane(x)
#else
#define TGP_WAVE_ALL(pred) (pred)
#define TGP_WAVE_UNIFORM_INT(x) (x)
#endif
TGP_HD bool same_bits(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __double_as_longlong(a) == __double_as_longlong(b);
#else
    int64_t x, y;
    __builtin_memcpy(&x, &a, 8);
    __builtin_memcpy(&y, &b, 8);
    return x == y;
#endif
}
TGP_HD bool same_bits(const Dual&, const Dual&) { return false; }
template <class T> struct steady_capable { static constexpr bool value = false; };
template <> struct steady_capable<double> { static constexpr bool value = true; };

TGP_HD void set_real(double& x, const double* v, const double*, int i) { x = v[i]; }
TGP_HD void set_real(Dual& x, const double* v, const double* d, int i) { x = Dual(v[i], d ? d[i] : 0.0); }
// element i of a tiled record and of its tangent record (same indexing)
TGP_HD void tile_real(double& x, const double* v, const double*, int64_t i) { x = v[i]; }
TGP_HD void tile_real(Dual& x, const double* v, const double* t, int64_t i) { x = Dual(v[i], t ? t[i] : 0.0); }

enum : uint32_t { kTileA = 1u, kTilea = 2u, kTileQ = 4u, kTileH = 8u, kTileh = 16u, kTileR = 32u, kTileDt = 64u };
// Largest state dimension whose per-step passes evaluate closed-form SDE transitions in registers (above it the transitions stay a tiled record)
constexpr int kSdeInKernelMaxD = 8;
// ... and up to this dimension by a build of the per-step passes of their own (tgp_inst_sde.hip: no run-time branch in the passes that read
// tiled records -- measured at d = 3, T = 1e7: the branch alone cost them 8 % -- and two waves per SIMD at d = 3)
constexpr int kSdeBuildMaxD = 4;

// offsets (in doubles) inside the transition record (A, a, Q) and the emission record (H row, h, R); bit 0 => size
TGP_HD int tile_offset_t(uint32_t mask, uint32_t bit, int d) {
    int off = 0;
    if (mask & kTileDt) return bit == 0u ? 1 : 0;      // the record is the step's tau alone
    if (bit == kTileA) return off;
    if (mask & kTileA) off += d * d;
    if (bit == kTilea) return off;
    if (mask & kTilea) off += d;
    if (bit == kTileQ) return off;
    if (mask & kTileQ) off += d * d;
    return off;
}
TGP_HD int tile_offset_e(uint32_t mask, uint32_t bit, int d) {
    int off = 0;
    if (bit == kTileH) return off;
    if (mask & kTileH) off += d;
    if (bit == kTileh) return off;
    if (mask & kTileh) off += 1;
    if (bit == kTileR) return off;
    if (mask & kTileR) off += 1;
    return off;
}

TGP_HD int64_t fs_index(int64_t c, int i, int k, int L0, int NS) {
    return ((((c >> 6) * L0 + i) * NS + k) << 6) + (c & 63);
}

// Processing step r = tproc * p + j (tproc: processing TIME index, j: observation inside the time step).
// Storage indices:  emission time index te, transition index ttr (Reverse: the previous step's), micro index tm.
TGP_HD int64_t time_index(const ModelView& mv, int64_t tproc) { return mv.ordering == 0 ? tproc : mv.Tt - 1 - tproc; }
TGP_HD int64_t trans_index(const ModelView& mv, int64_t tproc) { return mv.ordering == 0 ? tproc : mv.Tt - tproc; }
// micro storage index (y, per-step R stream, mean / var outputs) of processing step r = cc*L0 + i, L0 % p == 0
TGP_HD int64_t micro_index(const ModelView& mv, int64_t cc, int i, int L0) {
    if (mv.p == 1) return time_index(mv, cc * (int64_t)L0 + i);
    const int tl = i / mv.p, j = i - tl * mv.p;
    return time_index(mv, cc * (int64_t)(L0 / mv.p) + tl) * mv.p + j;
}

// Tiling of the per-step arrays of `raw` (reference layout, strides raw.s*). Transitions are stored at the
// processing time step that APPLIES them (Reverse: the previous storage index; the skipped predict at tproc = 0
// is zero-filled).
TGP_HD void tile_transition(const ModelView& raw, int d, uint32_t mask, int nc, int Lt, int64_t c, int tl, double* tile) {
    const int64_t tproc = c * (int64_t)Lt + tl;
    if (tproc >= raw.Tt || nc == 0) return;
    const int64_t tt = trans_index(raw, tproc);
    const bool pred = !(raw.ordering != 0 && tproc == 0);
    const int64_t base = fs_index(c, tl, 0, Lt, nc);
    int off = 0;
    if (mask & kTileA) { for (int k = 0; k < d * d; ++k) tile[base + (int64_t)(off + k) * 64] = pred ? raw.A[tt * raw.sA + k] : 0.0; off += d * d; }
    if (mask & kTilea) { for (int k = 0; k < d; ++k) tile[base + (int64_t)(off + k) * 64] = pred ? raw.a[tt * raw.sa + k] : 0.0; off += d; }
    if (mask & kTileQ) { for (int k = 0; k < d * d; ++k) tile[base + (int64_t)(off + k) * 64] = pred ? raw.Q[tt * raw.sQ + k] : 0.0; off += d * d; }
}
TGP_HD void tile_emission(const ModelView& raw, int d, uint32_t mask, int nc, int L0, int64_t c, int i, double* tile) {
    const int64_t r = c * (int64_t)L0 + i;
    if (r >= raw.T || nc == 0) return;
    const int tl = i / raw.p, j = i - tl * raw.p;
    const int64_t te = time_index(raw, c * (int64_t)(L0 / raw.p) + tl);
    const int64_t base = fs_index(c, i, 0, L0, nc);
    int off = 0;
    if (mask & kTileH) { for (int k = 0; k < d; ++k) tile[base + (int64_t)(off + k) * 64] = raw.H[te * raw.sH + j * d + k]; off += d; }
    if (mask & kTileh) { tile[base + (int64_t)off * 64] = raw.h[te * raw.sh + j]; off += 1; }
    if (mask & kTileR) { tile[base + (int64_t)off * 64] = raw.R[te * raw.sR + j]; off += 1; }
}

TGP_HD uint32_t tile_mask_of(const ModelView& raw) {
    return (raw.sA ? kTileA : 0u) | (raw.sa ? kTilea : 0u) | (raw.sQ ? kTileQ : 0u) | (raw.sH ? kTileH : 0u) |
           (raw.sh ? kTileh : 0u) | (raw.sR ? kTileR : 0u);
}

// Per-step SCALAR streams (y, per-step R, R_new in; mean/var out) are accessed through an IO object in
// groups of G consecutive steps. A lane's chunk is contiguous in time, so lane-wise access is strided by
// L0*8 bytes across the wave; the device IO (WaveIO, tgp_kernels.hpp) instead lets 8 lanes fetch / write
// one chunk's 64 contiguous bytes and transposes through wave-private LDS. DirectIO is the plain form
// (host emulation, and the reference semantics the staged form must reproduce). `tm` = micro storage index.
// Largest state dimension that gets the register-resident software prefetch (IO groups, smoother scratch, hoisted shared R).
constexpr int kPrefetchMaxD = 4;
// Largest state dimension with the stationary-covariance steps (tgp_chunk_body.inc): the two remembered steps cost 2 x (2 d^2 + d + 2)
// registers in pass 2; at d = 4 that build sits at the 512-register budget with spills.
constexpr int kSteadyMaxD = 3;

struct DirectIO {
    static constexpr int G = 8;
    const double* a0;  // y (or eps_e)
    const double* a1;  // per-step R (or R_new); only read when its stride is non-zero
    double* o0;
    double* o1;
    TGP_HD void begin(const ModelView&, int64_t, int, int) {}
    TGP_HD double in0(int64_t tm, int) const { return a0[tm]; }
    TGP_HD double in1(int64_t tm, int) const { return a1[tm]; }
    TGP_HD void out(int64_t tm, int, double x0, double x1) {
        o0[tm] = x0;
        if (o1) o1[tm] = x1;
    }
    TGP_HD void flush(const ModelView&, int64_t, int, int) {}
};

// Chunk bounds; lanes past the last chunk (c >= n0) get an empty range but still take part in the
// wave-cooperative IO.
TGP_HD void chunk_range(const ModelView& mv, int64_t c, int L0, int64_t& r0, int64_t& r1) {
    r0 = c * (int64_t)L0;
    if (r0 > mv.T) r0 = mv.T;
    r1 = r0 + L0 < mv.T ? r0 + L0 : mv.T;
}

struct FilterOut {
    double* m_out;  // [T][d]      filtering means      (MODE >= 1, may be null)
    double* P_out;  // [T][d*d]    filtering covariances
    double* fs;     // filtered-state scratch, fs_index layout (MODE 2)
    double* G_out;  // [T][d*d] [T][d] [T][d*d] materialised reverse model (MODE 2, may be null)
    double* g_out;
    double* L_out;
    double* xfin;   // MODE 3 of a Reverse-ordered prior: x0 of the posterior (the state after the last step's predict), packed
};

// The closed-form transition of one step (see ModelView::sde): tau >= 0 the gap to the previous time stamp, tau < 0 the first one.
template <int D> TGP_HD void sde_transition_impl(const double* q, int sde_first, double tau, double* A, double* Q) {
    if (tau < 0.0) {
        if (sde_first) {     // (A1, Q1) as the host evaluated them (kernel algebra decides each term's own dt_1, tgp_hip.h)
            TGP_UNROLL for (int k = 0; k < D * D; ++k) { A[k] = q[D + 3 * D * D + k]; Q[k] = q[D + 4 * D * D + k]; }
            return;
        }
        tau = 1.0;              // lti_sde.jl:139
    }
    double e[D];
    e[0] = ::exp(-q[0] * tau);
    TGP_UNROLL for (int i = 1; i < D; ++i) e[i] = (q[i] == q[i - 1]) ? e[i - 1] : ::exp(-q[i] * tau);     // (wave-uniform: one exp per block)
    const double t2 = tau * tau;
    TGP_UNROLL for (int j = 0; j < D; ++j) {
        TGP_UNROLL for (int i = 0; i < D; ++i) {
            const double c = ::fma(t2, q[D + D * D + i + j * D], ::fma(tau, q[D + i + j * D], i == j ? 1.0 : 0.0));
            A[i + j * D] = e[i] * c;
        }
    }
    // Q = Pinf - A Pinf A' (upper triangle, mirrored)
    const double* P = q + D + 2 * D * D;
    double AP[D * D];
    TGP_UNROLL for (int j = 0; j < D; ++j) {
        TGP_UNROLL for (int i = 0; i < D; ++i) {
            double acc = 0.0;
            TGP_UNROLL for (int k = 0; k < D; ++k) acc = ::fma(A[i + k * D], P[k + j * D], acc);
            AP[i + j * D] = acc;
        }
    }
    TGP_UNROLL for (int j = 0; j < D; ++j) {
        TGP_UNROLL for (int i = 0; i <= j; ++i) {
            double acc = 0.0;
            TGP_UNROLL for (int k = 0; k < D; ++k) acc = ::fma(AP[i + k * D], A[j + k * D], acc);
            const double v = P[i + j * D] - acc;
            Q[i + j * D] = v;
            Q[j + i * D] = v;
        }
    }
}
// (out-of-line from TGP_BIG_D on, like every other building block of that build: its temporaries must not join the caller's register budget)
template <int D> TGP_NOINLINE void sde_transition_out(const double* q, int sde_first, double tau, double* A, double* Q) { sde_transition_impl<D>(q, sde_first, tau, A, Q); }
template <int D> TGP_HD void sde_transition(const ModelView& mv, double tau, double* A, double* Q) {
    if constexpr (D >= TGP_BIG_D) { sde_transition_out<D>(mv.sde, mv.sde_first, tau, A, Q); } else { sde_transition_impl<D>(mv.sde, mv.sde_first, tau, A, Q); }
}
template <int D> TGP_HD void sde_transition(const ModelView&, double, Dual*, Dual*) {}   // (the gradient pass reads tiled tangents instead)

using real_t = double;
#include "tgp_chunk_body.inc"

namespace ad {
using real_t = Dual;
#include "tgp_chunk_body.inc"
}  // namespace ad

}  // namespace TGP_NS
