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

namespace tgp {

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
};

enum : uint32_t { kTileA = 1u, kTilea = 2u, kTileQ = 4u, kTileH = 8u, kTileh = 16u, kTileR = 32u };

// offsets (in doubles) inside the transition record (A, a, Q) and the emission record (H row, h, R); bit 0 => size
TGP_HD int tile_offset_t(uint32_t mask, uint32_t bit, int d) {
    int off = 0;
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

// Loads one processing step. LTI == true: A, a, Q, H, h are shared (hoisted once when p == 1; for p > 1 the
// observation row j is a wave-uniform load per step); only y and, when per-step, R stream (through the IO).
// LTI == false: per-step arrays come from the time-tiled records (coalesced rows), the others are shared.
template <int D, bool LTI> struct StepLoader {
    double A[D * D], a[D], Q[D * D], H[D], h, R, y;
    bool do_predict, is_missing;
    int64_t te, tm;   // time / micro storage index of the current step
    int j, tl;        // observation inside the time step, time step inside the chunk
    int oA, oa, oQ, oH, oh, oR;

    TGP_HD void init(const ModelView& mv) {
        const uint32_t m = LTI ? 0u : mv.tile_mask;
        if (!(m & kTileA)) { TGP_UNROLL for (int i = 0; i < D * D; ++i) A[i] = mv.A[i]; }
        if (!(m & kTileQ)) { TGP_UNROLL for (int i = 0; i < D * D; ++i) Q[i] = mv.Q[i]; }
        if (!(m & kTilea)) { TGP_UNROLL for (int i = 0; i < D; ++i) a[i] = mv.a[i]; }
        if (!(m & kTileH) && mv.p == 1) { TGP_UNROLL for (int i = 0; i < D; ++i) H[i] = mv.H[i]; }
        if (!(m & kTileh) && mv.p == 1) h = mv.h[0];
        if (!LTI) {
            oA = tile_offset_t(m, kTileA, D); oa = tile_offset_t(m, kTilea, D); oQ = tile_offset_t(m, kTileQ, D);
            oH = tile_offset_e(m, kTileH, D); oh = tile_offset_e(m, kTileh, D); oR = tile_offset_e(m, kTileR, D);
        }
    }
    // c = chunk, i = step inside the chunk (L0 % p == 0)
    TGP_HD void index(const ModelView& mv, int64_t c, int i, int L0) {
        int64_t tproc;
        if (mv.p == 1) {
            tl = i; j = 0;
            tproc = c * (int64_t)L0 + i;
        } else {
            tl = i / mv.p; j = i - tl * mv.p;
            tproc = c * (int64_t)(L0 / mv.p) + tl;
        }
        te = time_index(mv, tproc);
        tm = te * mv.p + j;
        do_predict = (j == 0) && !(mv.ordering != 0 && tproc == 0);
    }
    TGP_HD void load_transition(const ModelView& mv, int64_t c, int L0) {
        if (!LTI && do_predict && mv.nc_t > 0) {
            const double* q = mv.tile_t + fs_index(c, tl, 0, L0 / mv.p, mv.nc_t);
            if (mv.tile_mask & kTileA) { TGP_UNROLL for (int k = 0; k < D * D; ++k) A[k] = q[(oA + k) * 64]; }
            if (mv.tile_mask & kTileQ) { TGP_UNROLL for (int k = 0; k < D * D; ++k) Q[k] = q[(oQ + k) * 64]; }
            if (mv.tile_mask & kTilea) { TGP_UNROLL for (int k = 0; k < D; ++k) a[k] = q[(oa + k) * 64]; }
        }
    }
    TGP_HD void load_emission(const ModelView& mv, int64_t c, int i, int L0) {
        const uint32_t m = LTI ? 0u : mv.tile_mask;
        if (m & (kTileH | kTileh)) {
            const double* q = mv.tile_e + fs_index(c, i, 0, L0, mv.nc_e);
            if (m & kTileH) { TGP_UNROLL for (int k = 0; k < D; ++k) H[k] = q[(oH + k) * 64]; }
            if (m & kTileh) h = q[oh * 64];
        }
        if (mv.p > 1) {   // shared (Fill) emission with several rows: wave-uniform row j
            if (!(m & kTileH)) { TGP_UNROLL for (int k = 0; k < D; ++k) H[k] = mv.H[j * D + k]; }
            if (!(m & kTileh)) h = mv.h[j];
        }
    }
    // R of this step: tiled record (general layout) / shared / staged stream (LTI with per-step R)
    template <class IO> TGP_HD double load_R(const ModelView& mv, int64_t c, int i, int L0, const IO& io, int gi) const {
        if (!LTI && (mv.tile_mask & kTileR)) return mv.tile_e[fs_index(c, i, oR, L0, mv.nc_e)];
        return (mv.sR == 0) ? mv.R[j] : io.in1(tm, gi);
    }
    template <class IO> TGP_HD void load_obs(const ModelView& mv, int64_t c, int i, int L0, const IO& io, int gi) {
        R = load_R(mv, c, i, L0, io, gi);
        y = io.in0(tm, gi);
        is_missing = (mv.missing != nullptr) && (mv.missing[tm] != 0);
        if (is_missing) { y = 0.0; R = kLargeVar; }
    }
    // full step: i = step inside the chunk, gi = position inside the IO group
    template <class IO> TGP_HD void load(const ModelView& mv, int64_t c, int i, int L0, const IO& io, int gi) {
        index(mv, c, i, L0);
        load_transition(mv, c, L0);
        load_emission(mv, c, i, L0);
        load_obs(mv, c, i, L0, io, gi);
    }
};

// Whether the IO must stage mv.R: only the LTI family streams a per-step R through it.
template <bool LTI> TGP_HD bool io_stages_R(const ModelView& mv) { return LTI && mv.sR != 0; }

// ------------------------------------------------------------------------------------------ pass 1
// Chunk bounds; lanes past the last chunk (c >= n0) get an empty range but still take part in the
// wave-cooperative IO.
TGP_HD void chunk_range(const ModelView& mv, int64_t c, int L0, int64_t& r0, int64_t& r1) {
    r0 = c * (int64_t)L0;
    if (r0 > mv.T) r0 = mv.T;
    r1 = r0 + L0 < mv.T ? r0 + L0 : mv.T;
}

template <int D, bool LTI, class IO, typename Store>
TGP_HD void chunk_reduce_filter(const ModelView& mv, int64_t c, int L0, IO& io, Store st) {
    int64_t r0, r1;
    chunk_range(mv, c, L0, r0, r1);
    FElem<D> e;
    e.identity();
    StepLoader<D, LTI> sl;
    sl.init(mv);
    for (int g = 0; g < L0; g += IO::G) {
        io.begin(mv, c, g, L0);
        const int64_t rg = r0 + g;
        const int gend = (int)((r1 - rg) < IO::G ? (r1 - rg) : IO::G);
        for (int i = 0; i < gend; ++i) {
            sl.load(mv, c, g + i, L0, io, i);
            f_extend<D>(e, sl.do_predict, sl.A, sl.a, sl.Q, sl.H, sl.h, sl.R, sl.y);
        }
    }
    if (r1 > r0) store_felem<D>(e, st);
}

// ------------------------------------------------------------------------------------------ pass 2
struct FilterOut {
    double* m_out;  // [T][d]      filtering means      (MODE >= 1, may be null)
    double* P_out;  // [T][d*d]    filtering covariances
    double* fs;     // filtered-state scratch, fs_index layout (MODE 2)
    double* G_out;  // [T][d*d] [T][d] [T][d*d] materialised reverse model (MODE 2, may be null)
    double* g_out;
    double* L_out;
};

struct ChunkStats {
    double lml;
    double nmiss;
    int32_t bad;  // 1 => non-positive innovation variance / Cholesky failure inside this chunk
};

// MODE 0: logpdf only. MODE 1: + filtering distributions. MODE 2: smoother forward pass -- filtering-state
// scratch + the chunk's smoother element, composed step by step from the reference's own (jittered)
// invert_dynamics and stored through `rst`. MODE 3: materialise the posterior model: per-step
// invert_dynamics -> (G, g, L) outputs (lgssm.jl:215-221), no scratch, no element.
// (A chunk-level closed form of the smoother element from the chunk's filter element -- chunk_smoother_element,
// tgp_math.hpp -- needs no per-step work, but it is the EXACT smoother: it differs from the reference's
// 1e-10-jittered recursion by up to 2.5e-8 on the bench parametrisation, so it is not used.)
template <int D, bool LTI, int MODE, class IO, typename RStore>
TGP_HD ChunkStats chunk_apply_filter(const ModelView& mv, int64_t c, int L0, State<D>& x, const FilterOut& out, IO& io, RStore rst) {
    int64_t r0, r1;
    chunk_range(mv, c, L0, r0, r1);
    ChunkStats cs{0.0, 0.0, 0};
    StepLoader<D, LTI> sl;
    sl.init(mv);
    AElem<D> rev;
    if (MODE == 2) rev.identity();
    bool ok = true;
    for (int g = 0; g < L0; g += IO::G) {
      io.begin(mv, c, g, L0);
      const int64_t rg = r0 + g;
      const int gend = (int)((r1 - rg) < IO::G ? (r1 - rg) : IO::G);
      double sprod = 1.0, quad = 0.0;   // product of the group's innovation variances, sum of v^2 / S
      for (int gi = 0; gi < gend; ++gi) {
        const int64_t r = rg + gi;
        sl.load(mv, c, g + gi, L0, io, gi);
        if (MODE >= 2 && sl.do_predict) {
            double mf[D], Pf[D * D];
            copy_n<D>(x.m, mf);
            copy_n<D * D>(x.P, Pf);
            predict<D>(sl.A, sl.a, sl.Q, x.m, x.P);
            double G[D * D], g_[D], L[D * D];
            ok = invert_dynamics<D>(mf, Pf, x.m, x.P, sl.A, G, g_, L) && ok;
            if (MODE == 2) a_extend_right<D>(rev, G, g_, L);
            if (MODE == 3 && out.G_out) {
                TGP_UNROLL for (int i = 0; i < D * D; ++i) { out.G_out[sl.te * D * D + i] = G[i]; out.L_out[sl.te * D * D + i] = L[i]; }
                TGP_UNROLL for (int i = 0; i < D; ++i) out.g_out[sl.te * D + i] = g_[i];
            }
        } else if (sl.do_predict) {
            predict<D>(sl.A, sl.a, sl.Q, x.m, x.P);
        }
        double S;
        quad += update_scalar_nolog<D>(sl.H, sl.h, sl.R, sl.y, x.m, x.P, ok, S);
        sprod *= S;
        if (sprod > 1e100 || sprod < 1e-100) {   // keep the running product far from over/underflow
            cs.lml -= 0.5 * log(sprod);
            sprod = 1.0;
        }
        cs.nmiss += sl.is_missing ? 1.0 : 0.0;
        if (MODE >= 1 && out.m_out && sl.j == mv.p - 1) {
            TGP_UNROLL for (int i = 0; i < D; ++i) out.m_out[sl.te * D + i] = x.m[i];
        }
        if (MODE >= 1 && out.P_out && sl.j == mv.p - 1) {
            TGP_UNROLL for (int i = 0; i < D * D; ++i) out.P_out[sl.te * D * D + i] = x.P[i];
        }
        if (MODE == 2 && out.fs) {
            int i = (int)(r - r0);
            double* fs = out.fs;
            int L0_ = L0;
            store_state<D>(x, [=](int k, double v) { fs[fs_index(c, i, k, L0_, Dim<D>::NS)] = v; });
        }
      }
      // lml of the group: -(n log 2pi + log prod S + sum v^2/S) / 2   (lgc.jl:254 summed over the group)
      if (gend > 0) cs.lml -= 0.5 * (gend * kLog2Pi + log(sprod) + quad);
    }
    if (MODE == 2 && r1 > r0) store_aelem<D>(rev, rst);
    cs.bad = ok ? 0 : 1;
    return cs;
}

// ------------------------------------------------------------------------------------------ pass 3
// RTS smoother on chunk c. `xs` = smoothed state at the chunk's LAST step; `carry` = filtered state
// just before the chunk's first step. Emits N(H x + h, H P H' + Rnew) for every step (lgssm.jl:111-115
// on the posterior model with replace_observation_noise_cov, missings.jl:35-41).
template <int D, bool LTI, class IO>
TGP_HD int chunk_smooth(const ModelView& mv, int64_t c, int L0, State<D>& xs, const State<D>& carry, const double* fs,
                        int64_t sRn, IO& io) {
    int64_t r0, r1;
    chunk_range(mv, c, L0, r0, r1);
    StepLoader<D, LTI> sl;
    sl.init(mv);
    bool ok = true;

    for (int g = ((L0 - 1) / IO::G) * IO::G; g >= 0; g -= IO::G) {
        io.begin(mv, c, g, L0);   // stages R_new (io.a1) when it is per-step
        const int64_t rg = r0 + g;
        const int gend = (int)((r1 - rg) < IO::G ? (r1 - rg) : IO::G);
        for (int gi = gend - 1; gi >= 0; --gi) {
            const int64_t r = rg + gi;
            sl.index(mv, c, g + gi, L0);
            sl.load_transition(mv, c, L0);
            sl.load_emission(mv, c, g + gi, L0);
            double mean, var;
            emit_scalar<D>(sl.H, sl.h, (sRn == 0) ? io.a1[sl.j] : io.in1(sl.tm, gi), xs.m, xs.P, mean, var);
            io.out(sl.tm, gi, mean, var);
            if (!sl.do_predict) continue;   // inside a time step (or the skipped first predict): the state does not move
            State<D> xf;  // filtered state before this time step
            if (r == r0) {
                xf = carry;
            } else {
                int i = (int)(r - r0) - 1;
                int L0_ = L0;
                load_state<D>(xf, [=](int k) { return fs[fs_index(c, i, k, L0_, Dim<D>::NS)]; });
            }
            double mp[D], Pp[D * D];
            copy_n<D>(xf.m, mp);
            copy_n<D * D>(xf.P, Pp);
            predict<D>(sl.A, sl.a, sl.Q, mp, Pp);
            double G[D * D], g_[D], L[D * D];
            ok = invert_dynamics<D>(xf.m, xf.P, mp, Pp, sl.A, G, g_, L) && ok;
            predict<D>(G, g_, L, xs.m, xs.P);
        }
        io.flush(mv, c, g, L0);
    }
    return ok ? 0 : 1;
}

// ------------------------------------------------------------------------------------------ affine passes
// Prior marginals (COV, !RAND): x' = A x + a, P' = A P A' + Q; emits N(H x + h, H P H' + R).
// rand (RAND, !COV): x' = A x + a + chol(Q + 1e-9 I).U' eps_t ; y = H x' + h + sqrt(R) eps_e.
template <int D> TGP_HD bool noise_factor(const double* Q, double* Lq) {  // lower factor, column-major
    double Qj[D * D], U[D * D];
    copy_n<D * D>(Q, Qj);
    TGP_UNROLL for (int i = 0; i < D; ++i) Qj[i + i * D] += 1e-9;  // lgc.jl:86
    bool ok = chol_upper<D>(Qj, U);
    TGP_UNROLL for (int j = 0; j < D; ++j) TGP_UNROLL for (int i = 0; i < D; ++i) Lq[i + j * D] = U[j + i * D];
    return ok;
}

template <int D, bool LTI, bool RAND, typename Store>
TGP_HD int chunk_reduce_affine(const ModelView& mv, int64_t c, int L0, const double* eps_t, Store st) {
    int64_t r0, r1;
    chunk_range(mv, c, L0, r0, r1);
    AElem<D> e;
    e.identity();
    StepLoader<D, LTI> sl;
    sl.init(mv);
    double Lq[D * D];
    bool ok = true;
    if (RAND && LTI) ok = noise_factor<D>(sl.Q, Lq);
    for (int64_t r = r0; r < r1; ++r) {
        sl.index(mv, c, (int)(r - r0), L0);
        sl.load_transition(mv, c, L0);
        if (!sl.do_predict) continue;
        if (RAND) {
            if (!LTI) ok = noise_factor<D>(sl.Q, Lq) && ok;
            double cvec[D];
            const double* ep = eps_t + trans_index(mv, c * (int64_t)(L0 / mv.p) + sl.tl) * D;
            TGP_UNROLL for (int i = 0; i < D; ++i) {
                double acc = sl.a[i];
                TGP_UNROLL for (int k = 0; k <= i; ++k) acc = fma(Lq[i + k * D], ep[k], acc);
                cvec[i] = acc;
            }
            a_extend<D, false>(e, sl.A, cvec, sl.Q);
        } else {
            a_extend<D, true>(e, sl.A, sl.a, sl.Q);
        }
    }
    store_aelem<D>(e, st);
    return ok ? 0 : 1;
}

template <int D, bool LTI, bool RAND, class IO>
TGP_HD int chunk_apply_affine(const ModelView& mv, int64_t c, int L0, State<D>& x, const double* eps_t, IO& io) {
    int64_t r0, r1;
    chunk_range(mv, c, L0, r0, r1);
    StepLoader<D, LTI> sl;
    sl.init(mv);
    double Lq[D * D];
    bool ok = true;
    if (RAND && LTI) ok = noise_factor<D>(sl.Q, Lq);
    for (int g = 0; g < L0; g += IO::G) {
      io.begin(mv, c, g, L0);   // RAND: stages eps_e (io.a0); per-step R (io.a1)
      const int64_t rg = r0 + g;
      const int gend = (int)((r1 - rg) < IO::G ? (r1 - rg) : IO::G);
      for (int gi = 0; gi < gend; ++gi) {
        const int64_t r = rg + gi;
        sl.index(mv, c, g + gi, L0);
        sl.load_transition(mv, c, L0);
        sl.load_emission(mv, c, g + gi, L0);
        const double R = sl.load_R(mv, c, g + gi, L0, io, gi);
        if (sl.do_predict) {
            if (RAND) {
                if (!LTI) ok = noise_factor<D>(sl.Q, Lq) && ok;
                const double* ep = eps_t + trans_index(mv, c * (int64_t)(L0 / mv.p) + sl.tl) * D;
                double xn[D];
                mat_vec<D>(sl.A, x.m, xn);
                TGP_UNROLL for (int i = 0; i < D; ++i) {
                    double nz = 0.0;
                    TGP_UNROLL for (int k = 0; k <= i; ++k) nz = fma(Lq[i + k * D], ep[k], nz);
                    x.m[i] = (xn[i] + sl.a[i]) + nz;
                }
            } else {
                predict<D>(sl.A, sl.a, sl.Q, x.m, x.P);
            }
        }
        if (RAND) {
            double yy = 0.0;
            TGP_UNROLL for (int i = 0; i < D; ++i) yy = fma(sl.H[i], x.m[i], yy);
            // scalar emission: sqrt(R) eps (lgc.jl:241-243); vector emission with diagonal R: chol(R + 1e-9 I).U' eps (lgc.jl:84-87)
            io.out(sl.tm, gi, (yy + sl.h) + sqrt(mv.small_out ? R + 1e-9 : R) * io.in0(sl.tm, gi), 0.0);
        } else {
            double mean, var;
            emit_scalar<D>(sl.H, sl.h, R, x.m, x.P, mean, var);
            io.out(sl.tm, gi, mean, var);
        }
      }
      io.flush(mv, c, g, L0);
    }
    return ok ? 0 : 1;
}

}  // namespace tgp
