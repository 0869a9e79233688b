// Dense large-state Kalman engine for gfx950 (see tgp_dense.hpp). One time step = a short fixed chain of kernels on the
// handle's stream; every O(d^3) / O(p d^2) contraction runs on v_mfma_f64_16x16x4_f64.
//
// Layouts (all fp64, device): dimensions are padded to multiples of 16 (Dp >= d, Pq >= p) with zeros (R padded with ones), which
// leaves every quantity of the recursion unchanged (the padded rows / columns of P stay zero, the padded block of S is I).
//   Ak  [Dp x Dp]  column-major A (A[i][k] at Ak[i + k Dp])        Qc [Dp x Dp] column-major Q
//   Hk  [Dp][Pq]   H transposed: H[i][k] at Hk[k Pq + i]           av [Dp], hv [Pq], Rv [Pq]
//   P, Pp, T1 [Dp x Dp] column-major; V [Pq x (Dp + 16)] column-major (V[i][k] at V[i + k Pq]; column Dp holds the residual r)
//   S, L [Pq x Pq] column-major;  Bm [Pq][Dp + 16] row-major (B[k][i] at Bm[k ldB + i]; column Dp holds alpha)
// Every GEMM operand is therefore "k-major": Aop[i][k] at A[k lda + i], Bop[k][j] at B[k ldb + j] -- one coalesced row of the
// free index per k -- and every output is written with its first index contiguous.
#include "tgp_dense.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/tgp_hip.h"

namespace tgp_dense {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr double kLargeVar = 1e15;                 // missings.jl:43
constexpr double kLog2Pi = 1.8378770664093454836;  // log(2 pi)
constexpr size_t kOperandSlack = 64;                // doubles of slack behind every GEMM operand buffer (see dk_gemm)

// ------------------------------------------------------------------------------------------------ riders
// Small matrix-vector products that ride along a GEMM launch as extra workgroups (one kernel boundary less per product).
struct GemvArgs {
    int mode = 0;            // 0 none; 1 out = add + Mx x; 2 residual r = y - h - Mx x (+ missing count); 3 m = add + Mx x and lml_t
    const double* Mx = nullptr;  // Mx[i][k] at Mx[i + k ld]
    int64_t ld = 0;
    int n = 0, K = 0;        // rows (padded), inner length
    const double* x = nullptr;
    int64_t xs = 1;
    const double* add = nullptr;
    double* out = nullptr;
    double* out2 = nullptr;  // unpadded copy (first n2 rows), nullable
    int n2 = 0;
    const double* y = nullptr;       // mode 2
    const uint8_t* mask = nullptr;   // mode 2 (nullable)
    const double* hh = nullptr;      // mode 2
    int p = 0;                       // true observation count
    double* scal = nullptr;          // [0] logdet S (written by dk_chol), [1] missing elements of this step, [2] not-PD flag
    double* stats = nullptr;         // the call's result8
    int64_t tstep = 0;
};

// workgroup barrier that only waits for this wave's LDS traffic (a plain __syncthreads() also drains every outstanding global
// load / store, which would serialise the prefetches that are deliberately kept in flight across it)
__device__ inline void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 16 rows per workgroup: thread = (row = tid & 15, K slice = tid >> 4 of 32), four independent accumulators per thread so that
// the loads of a slice are in flight together; the 32 partial sums of a row are added in a fixed order.
__device__ inline void gemv_rider(const GemvArgs& v, int rb, double* lds) {
    const int tid = threadIdx.x;
    const int row = tid & 15, ks = tid >> 4;
    const int r = rb * 16 + row;
    const int kc = (v.K + 31) / 32;
    const int k0 = ks * kc, k1 = min(v.K, k0 + kc);
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (r < v.n) {
        const double* mp = v.Mx + r;
        int k = k0;
        for (; k + 4 <= k1; k += 4) {
            s0 += mp[(int64_t)k * v.ld] * v.x[(int64_t)k * v.xs];
            s1 += mp[(int64_t)(k + 1) * v.ld] * v.x[(int64_t)(k + 1) * v.xs];
            s2 += mp[(int64_t)(k + 2) * v.ld] * v.x[(int64_t)(k + 2) * v.xs];
            s3 += mp[(int64_t)(k + 3) * v.ld] * v.x[(int64_t)(k + 3) * v.xs];
        }
        for (; k < k1; ++k) s0 += mp[(int64_t)k * v.ld] * v.x[(int64_t)k * v.xs];
    }
    lds[ks * 16 + row] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (tid < 16 && r < v.n) {
        double tot = 0.0;
#pragma unroll
        for (int u = 0; u < 32; ++u) tot += lds[u * 16 + row];
        double o;
        if (v.mode == 2) {
            double yy = 0.0, hv = 0.0;
            if (r < v.p) {
                const bool miss = v.mask != nullptr && v.mask[r] != 0;
                yy = miss ? 0.0 : v.y[r];
                hv = v.hh[r];
            }
            o = r < v.p ? (yy - hv - tot) : 0.0;
        } else {
            o = tot + (v.add ? v.add[r] : 0.0);
        }
        v.out[r] = o;
        if (v.out2 && r < v.n2) v.out2[r] = o;
    }
    if (rb != 0) return;
    if (v.mode == 2) {   // count this step's missing elements
        __syncthreads();
        int* cnt = reinterpret_cast<int*>(lds);
        if (tid == 0) *cnt = 0;
        __syncthreads();
        int c = 0;
        if (v.mask)
            for (int i = tid; i < v.p; i += blockDim.x) c += v.mask[i] != 0;
        if (c) atomicAdd(cnt, c);
        __syncthreads();
        if (tid == 0) v.scal[1] = (double)*cnt;
    } else if (v.mode == 3) {   // lml_t = -(p log 2pi + logdet S + alpha' alpha) / 2 (+ missing-data compensation, missings.jl:45-53)
        __syncthreads();
        double q = 0.0;
        for (int k = tid; k < v.K; k += blockDim.x) {
            const double a = v.x[(int64_t)k * v.xs];
            q += a * a;
        }
        lds[tid] = q;
        __syncthreads();
        for (int st = blockDim.x / 2; st > 0; st >>= 1) {
            if (tid < st) lds[tid] += lds[tid + st];
            __syncthreads();
        }
        if (tid == 0) {
            const double nm = v.scal[1];
            const double lml = -0.5 * ((double)v.p * kLog2Pi + v.scal[0] + lds[0]) + nm * 0.5 * (kLog2Pi + log(kLargeVar));
            v.stats[0] += lml;
            v.stats[1] += nm;
            if (v.scal[2] != 0.0 && v.stats[2] == 0.0) v.stats[2] = (double)(v.tstep + 1);
        }
    }
}

// ------------------------------------------------------------------------------------------------ GEMM on fp64 MFMA
struct GemmArgs {
    const double* A = nullptr;   // Aop[i][k] at A[k lda + i]
    int64_t lda = 0;
    const double* B = nullptr;   // Bop[k][j] at B[k ldb + j]
    int64_t ldb = 0;
    double* C = nullptr;         // C[i][j] at C[i + j ldc]
    int64_t ldc = 0;
    const double* E = nullptr;   // C = E + sign * (Aop Bop) (+ diag)
    int64_t lde = 0;
    double sign = 1.0;
    const double* diag = nullptr;    // + diag (missing -> 1e15, rows >= ndiag -> 1): the S = V H' + R epilogue
    const uint8_t* dmask = nullptr;
    int ndiag = 0;
    double* C2 = nullptr;        // optional second (unpadded) copy of the result: C2[i + j ldc2], i < M2, j < N2
    int64_t ldc2 = 0;
    int M2 = 0, N2 = 0;
    int M = 0, N = 0, K = 0;
    int nblk_tiles = 0;
    GemvArgs v;
};

__device__ inline d4 mfma_f64(double a, double b, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

template <int TM, int TN> struct GemmCfg {
    static constexpr int NT = 512;
    static constexpr int BM = TM * 16, BN = TN * 16;
    static constexpr int NTILE = TM * TN;
    static constexpr int PF = 24 / (TM + TN) < 2 ? 2 : 24 / (TM + TN);   // k-steps per register buffer (two buffers)
    static constexpr int RED = 4 * NTILE * 256;
    static constexpr int LDS_DOUBLES = RED > 512 ? RED : 512;
    static constexpr size_t LDS_BYTES = (size_t)LDS_DOUBLES * sizeof(double);
};

// One workgroup (8 waves) owns a (16 TM) x (16 TN) block of C and its 8 waves split K (a contiguous eighth each), so a wave
// carries TM*TN accumulators and streams its own operand fragments STRAIGHT from global memory into MFMA operand registers:
// with the k-major layouts a fragment is four 128-byte rows (lanes 0-15 = 16 consecutive values of the free index, lanes
// 16-31 the next k, ...), no operand is used by two waves, so LDS staging would add a copy, a barrier per slab and nothing
// else. Two register buffers of PF k-steps: the loads of group g+1 are in flight while group g is multiplied. The 8 partial
// sums are then reduced through LDS in a fixed order (bit-reproducible).
template <int TM, int TN> __global__ __launch_bounds__(512) void dk_gemm(const GemmArgs g) {
    using Cfg = GemmCfg<TM, TN>;
    constexpr int BM = Cfg::BM, BN = Cfg::BN, NTILE = Cfg::NTILE, PF = Cfg::PF;
    extern __shared__ double lds[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    if ((int)blockIdx.x >= g.nblk_tiles) {
        gemv_rider(g.v, (int)blockIdx.x - g.nblk_tiles, lds);
        return;
    }
    const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN - 1) / BN, ntiles = ntm * ntn;
    // XCD-aware tile order: workgroup b runs on XCD b % 8; each XCD gets a contiguous run of tiles, and tiles are numbered in
    // bands of 4 tile rows, column by column inside a band, so that an XCD's run is a compact (4 rows x few columns) block.
    const int per = g.nblk_tiles >> 3;
    const int tl = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
    if (tl >= ntiles) return;
    const int band = tl / (4 * ntn), rem = tl - band * 4 * ntn;
    const int rows_in_band = min(4, ntm - band * 4);
    const int tn = rem / rows_in_band, tm = band * 4 + rem % rows_in_band;
    const int i0 = tm * BM, j0 = tn * BN;

    d4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = d4{0.0, 0.0, 0.0, 0.0};

    const int nks = g.K >> 2;                       // MFMA k-steps (K is a multiple of 16)
    const int kper = (nks + 7) >> 3;
    const int ks_begin = w * kper, ks_end = min(nks, ks_begin + kper);
    const double* Ap = g.A + i0 + (lane & 15) + (int64_t)(lane >> 4) * g.lda;
    const double* Bp = g.B + j0 + (lane & 15) + (int64_t)(lane >> 4) * g.ldb;
    // No predicates on the loads (a predicated load becomes a branch with a full wait behind it): tiles of this block that lie
    // beyond M / N read whatever follows -- every operand buffer carries kOperandSlack doubles of slack -- into accumulators
    // that are never stored; the k range is cut into whole groups of PF k-steps plus single steps.
    double fa0[PF][TM], fb0[PF][TN], fa1[PF][TM], fb1[PF][TN];
    auto load = [&](double (&fa)[PF][TM], double (&fb)[PF][TN], int ks0) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
#pragma unroll
            for (int a = 0; a < TM; ++a) fa[u][a] = Ap[(int64_t)(ks0 + u) * 4 * g.lda + a * 16];
#pragma unroll
            for (int b = 0; b < TN; ++b) fb[u][b] = Bp[(int64_t)(ks0 + u) * 4 * g.ldb + b * 16];
        }
    };
    auto compute = [&](double (&fa)[PF][TM], double (&fb)[PF][TN]) {
#pragma unroll
        for (int u = 0; u < PF; ++u)
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) acc[a][b] = mfma_f64(fb[u][b], fa[u][a], acc[a][b]);   // D'[j][i]: lanes run along i
    };
    const int nst = max(0, ks_end - ks_begin), ngroups = nst / PF;
    if (ngroups > 0) load(fa0, fb0, ks_begin);
    for (int gi = 0; gi < ngroups; gi += 2) {
        if (gi + 1 < ngroups) load(fa1, fb1, ks_begin + (gi + 1) * PF);
        compute(fa0, fb0);
        if (gi + 1 < ngroups) {
            if (gi + 2 < ngroups) load(fa0, fb0, ks_begin + (gi + 2) * PF);
            compute(fa1, fb1);
        }
    }
    for (int ks = ks_begin + ngroups * PF; ks < ks_end; ++ks) {
        double ta[TM], tb[TN];
#pragma unroll
        for (int a = 0; a < TM; ++a) ta[a] = Ap[(int64_t)ks * 4 * g.lda + a * 16];
#pragma unroll
        for (int b = 0; b < TN; ++b) tb[b] = Bp[(int64_t)ks * 4 * g.ldb + b * 16];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) acc[a][b] = mfma_f64(tb[b], ta[a], acc[a][b]);
    }
    // ---- split-K reduction: waves 4..7 -> LDS, waves 0..3 add their partner and publish, then every wave finishes tiles
    double* red = lds;
    if (w >= 4) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[(((w - 4) * NTILE + a * TN + b) * 4 + r) * 64 + lane] = acc[a][b][r];
    }
    lds_barrier();
    if (w < 4) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double* q = &red[((w * NTILE + a * TN + b) * 4 + r) * 64 + lane];
                    *q = acc[a][b][r] + *q;
                }
    }
    lds_barrier();
    for (int q = w; q < NTILE; q += 8) {
        const int a = q / TN, b = q - a * TN;
        const int i = i0 + a * 16 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = j0 + b * 16 + (lane >> 4) + 4 * r;
            const double s01 = red[((0 * NTILE + q) * 4 + r) * 64 + lane] + red[((1 * NTILE + q) * 4 + r) * 64 + lane];
            const double s23 = red[((2 * NTILE + q) * 4 + r) * 64 + lane] + red[((3 * NTILE + q) * 4 + r) * 64 + lane];
            double val = g.sign * (s01 + s23);
            if (i < g.M && j < g.N) {
                if (g.E) val += g.E[i + (int64_t)j * g.lde];
                if (g.diag && i == j) val += i < g.ndiag ? ((g.dmask && g.dmask[i]) ? kLargeVar : g.diag[i]) : 1.0;
                g.C[i + (int64_t)j * g.ldc] = val;
                if (g.C2 && i < g.M2 && j < g.N2) g.C2[i + (int64_t)j * g.ldc2] = val;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ Cholesky of S (n <= 256)
// lane J (< 16, a compile-time constant after unrolling) of the wave to the lanes 0 .. 31: row_newbcast:J within the first row of 16 lanes, then
// row_bcast:15 (lane 15 of a row to the next row) into the second -- DPP moves (as fast as a v_readlane pair here, no faster: see dk_chol)
__device__ __forceinline__ double first_row_lane(double v, int j) {
    int lo = __double2loint(v), hi = __double2hiint(v);
#define DK_BC(J)                                                                   \
    case J:                                                                        \
        lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + J, 0xF, 0xF, true);        \
        hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + J, 0xF, 0xF, true);        \
        break;
    switch (j) {
        DK_BC(0) DK_BC(1) DK_BC(2) DK_BC(3) DK_BC(4) DK_BC(5) DK_BC(6) DK_BC(7) DK_BC(8) DK_BC(9) DK_BC(10) DK_BC(11) DK_BC(12) DK_BC(13) DK_BC(14) DK_BC(15)
    }
#undef DK_BC
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x142, 0xA, 0xF, false);      // (rows 1 and 3 take over; the others keep their own)
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x142, 0xA, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ inline double rdlane(double v, int l) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, l);
    hi = __builtin_amdgcn_readlane(hi, l);
    return __hiloint2double(hi, lo);
}

#include "tgp_dense_fused.hpp"

// One workgroup of 16 waves factorises S = L L' (right-looking, 16-wide panels). Waves 1..15 keep the lower 16 x 16 tiles in
// registers for the whole factorisation, TRANSPOSED in the MFMA accumulator layout (lane l, register r holds
// T[l & 15][(l >> 4) + 4 r]) -- in that layout a tile is directly the B operand of the MFMA that solves it against the panel's
// diagonal block, the accumulator of its rank-16 updates, and (once solved) BOTH operands of the update of the diagonal tile
// to its right. Wave 0 owns nothing: it is the CHAIN wave that factorises and inverts the 16 x 16 diagonal tile of panel kb
// while the other waves apply panel kb-1 to everything right of it. Per panel:
//   barrier 1 | wave 0: L_kk, W = L_kk^-1 from `dtile`           | waves 1-15: rank-16 update with panel kb-1 (columns >= kb)
//   barrier 2 | owners of (i, kb): L_ik' = W T_ik' (4 MFMAs) -> panel buffer (LDS, [tile][k][row]) and global L; the owner of
//             | (kb+1, kb) also owns (kb+1, kb+1) -- adjacent slots -- updates it straight from its registers and hands it
//             | to wave 0 through `dtile`
// The tile -> (wave, slot) table comes from the host (chol_slot_table). L (lower, column-major), the inverses of the diagonal
// tiles (Dinv[b][k][row], what dk_trsm multiplies with) and log det S go to global memory.
constexpr int kCholMaxTiles = 16;   // n <= 256
constexpr int kCholWorkers = 15;     // waves 1..15 own tiles, wave 0 is the chain wave
constexpr int kCholSlots = 10;       // ceil(136 / 15)

__device__ inline double rsqrt_fast(double x) {
    // v_rsq_f64 (~2^-26) + one third-order correction: relative error ~1e-16 (the reference takes sqrt and divides)
    const double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-x * y, y, 1.0);
    return fma(y * e, fma(0.375, e, 0.5), y);
}

#ifdef DK_TRACE
#define DK_STAMP(slot) do { if (lane == 0 && w < 2) trace[((kb) * 4 + (slot)) * 2 + (slot >= 2 ? 1 : 0)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define DK_STAMP(slot) do { } while (0)
#endif
__global__ __launch_bounds__(1024) void dk_chol(const double* __restrict__ S, int64_t ldS, int n, double jitter, const int* __restrict__ slots,
                                                 double* __restrict__ L, int64_t ldL, double* __restrict__ Dinv, double* __restrict__ scal
#ifdef DK_TRACE
                                                 , long long* __restrict__ trace
#endif
                                                 ) {
    __shared__ double pan[3][kCholMaxTiles * 256];
    __shared__ double dtile[256];
    __shared__ double winv[2][256];
    __shared__ double dg[kCholMaxTiles * 16];
    __shared__ int flag_w, flag_d, arrived, bad;   // monotonic: panels published by the chain wave / diagonal tiles handed over / worker arrivals
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt = n / 16;
    if (tid == 0) {
        bad = 0;
        flag_w = 0;
        flag_d = 0;
        arrived = 0;
    }
    __syncthreads();
    auto wait_ge = [&](int* f, int target) __attribute__((always_inline)) {
        while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(2);
    };
    auto publish = [&](int* f, int value) __attribute__((always_inline)) {   // all of this wave's LDS writes first, then the flag
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_store(f, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    int tt[kCholSlots];   // owned tile of slot s: (i << 4) | j, or -1
#define TI(s) (tt[s] >> 4)
#define TJ(s) (tt[s] < 0 ? -1 : (tt[s] & 15))
    d4 tile[kCholSlots];
#pragma unroll
    for (int s = 0; s < kCholSlots; ++s) {
        tt[s] = w > 0 ? slots[(w - 1) * kCholSlots + s] : -1;
        if (tt[s] >= 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                tile[s][r] = S[(TI(s) * 16 + (lane & 15)) + (int64_t)(TJ(s) * 16 + (lane >> 4) + 4 * r) * ldS];
                if (TI(s) == TJ(s) && (lane & 15) == (lane >> 4) + 4 * r) tile[s][r] += jitter;   // S + jitter I (lgssm.jl:235)
            }
            if (tt[s] == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) dtile[r * 64 + lane] = tile[s][r];
                publish(&flag_d, 1);
            }
        } else {
            tile[s] = d4{0.0, 0.0, 0.0, 0.0};
        }
    }
    if (w == 0) {
        // ---------------------------------------------------------------- chain wave: factorise + invert the diagonal tiles
        __builtin_amdgcn_s_setprio(3);
        const int idx = lane & 15;
        const bool is_w = (lane & 16) != 0;
        (void)idx;
        (void)is_w;
        int notpd = 0;
        for (int kb = 0; kb < nt; ++kb) {
            wait_ge(&flag_d, kb + 1);
            DK_STAMP(0);
#ifndef DK_CHAIN_SCALAR
            // Blocked by 4 columns, the tile in the MFMA accumulator layout it arrives in (t[r], lane l: T[row l & 15][col (l >> 4) + 4 r]):
            // REGISTER p of that layout IS the panel of columns 4p .. 4p + 3 as an MFMA operand (lane l: row l & 15, k = l >> 4) -- both
            // operands of the rank-4 update of the columns to its right, and the B operand of W_pp X' that turns the panel into L's columns.
            // Per panel: the 4 x 4 diagonal block to every lane (one LDS round trip), its Cholesky factor and inverse as uniform arithmetic
            // (four v_rsq chains), then four MFMAs: the panel, the trailing update, and the same two eliminations applied to W = L^-1
            // (kept in the D layout: w[r], lane l: W[(l >> 4) + 4 r][l & 15], from the identity).  Measured (scripts/dense_kbench.hip): 4.6K
            // cycles per tile with the workers idle, 5.2 - 7.7K beside their MFMAs, where the scalar recurrence over all 16 columns (below,
            // -DDK_CHAIN_SCALAR: 120 broadcast-and-update pairs per tile) takes 5.8K / 7 - 10.6K; the factorisation 68.7 -> 65.4 us -- the
            // workers' solve + urgent update + hand-over between two diagonal tiles (2 - 8K cycles) is now as long as the chain.
            d4 t, wv;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                t[r] = dtile[r * 64 + lane];
                wv[r] = ((lane >> 4) + 4 * r == (lane & 15)) ? 1.0 : 0.0;
            }
            const int row = lane & 15, grp = lane >> 4;
            double* sx = dtile;      // (free until the next diagonal tile is handed over, which waits for this panel's flag)
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const double x = t[p];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                sx[lane] = x;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const double* q = sx + 4 * p;      // D[a][b] = T[4p + a][4p + b] = sx[16 b + 4p + a]
                const double D00 = q[0], D10 = q[1], D20 = q[2], D30 = q[3], D11 = q[17], D21 = q[18], D31 = q[19], D22 = q[34], D32 = q[35], D33 = q[51];
                // Cholesky of the 4 x 4 block (1 / L[a][a] = rsqrt(pivot)) and its inverse, the same in every lane
                const double r0 = rsqrt_fast(D00), l10 = D10 * r0, l20 = D20 * r0, l30 = D30 * r0;
                const double p1 = fma(-l10, l10, D11), r1 = rsqrt_fast(p1), l21 = fma(-l20, l10, D21) * r1, l31 = fma(-l30, l10, D31) * r1;
                const double p2 = fma(-l21, l21, fma(-l20, l20, D22)), r2 = rsqrt_fast(p2), l32 = fma(-l31, l21, fma(-l30, l20, D32)) * r2;
                const double p3 = fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, D33))), r3 = rsqrt_fast(p3);
                notpd |= !(D00 > 0.0) | !(p1 > 0.0) | !(p2 > 0.0) | !(p3 > 0.0);
                const double w10 = -(l10 * r0) * r1, w21 = -(l21 * r1) * r2, w32 = -(l32 * r2) * r3;
                const double w20 = -fma(l21, w10, l20 * r0) * r2, w31 = -fma(l32, w21, l31 * r1) * r3;
                const double w30 = -fma(l32, w20, fma(l31, w10, l30 * r0)) * r3;
                // W_pp as the A operand of a 16 x 16 x 4 product: lane l holds A[I = l & 15][k = l >> 4], zero outside the 4 x 4 lower triangle
                double wp = 0.0;
                {
                    const double c0 = row == 0 ? r0 : (row == 1 ? w10 : (row == 2 ? w20 : w30));
                    const double c1 = row == 1 ? r1 : (row == 2 ? w21 : w31);
                    const double c2 = row == 2 ? r2 : w32;
                    const double v = grp == 0 ? c0 : (grp == 1 ? c1 : (grp == 2 ? c2 : r3));
                    wp = (row < 4 && grp <= row) ? v : 0.0;
                }
                // the panel: L[:, 4p + g] = X W_pp' -- as (W_pp X')', which lands in register 0 in the operand layout again
                const d4 zero4 = d4{0.0, 0.0, 0.0, 0.0};
                const double lp = mfma_f64(wp, x, zero4)[0];
                const int col = 4 * p + grp;
                const double lm = row >= col ? lp : 0.0;            // L is lower triangular (rows above: what earlier updates left there)
                t[p] = lm;
                if (p < 3) {
                    const d4 u = mfma_f64(-lm, lm, t);                 // T[:, > 4p + 3] -= L_p L_p'
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (r > p) t[r] = u[r];
                }
                // W <- L_p^-1 W: its rows 4p .. 4p + 3 times W_pp, then the rows below minus L[below, panel] times those
                wv[p] = mfma_f64(wp, wv[p], zero4)[0];
                if (p < 3) {
                    const double lb = row > 4 * p + 3 ? lp : 0.0;
                    wv = mfma_f64(-lb, wv[p], wv);
                }
                if (lane < 4) dg[kb * 16 + 4 * p + lane] = lane == 0 ? D00 : (lane == 1 ? p1 : (lane == 2 ? p2 : p3));
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                L[(kb * 16 + row) + (int64_t)(kb * 16 + grp + 4 * r) * ldL] = t[r];
                winv[kb & 1][row * 16 + grp + 4 * r] = wv[r];      // W[k = grp + 4 r][col = row] at [k' = col][row' = k]
                Dinv[kb * 256 + row * 16 + grp + 4 * r] = wv[r];
            }
#else
            // Lanes 0-15 hold one ROW of the tile each (z[k] = T[row][k]); lanes 16-31 hold one COLUMN of W = L_kk^-1 each
            // (z[k] = W[k][col], starting from the identity): both obey the same recurrence z[c] *= 1 / L[c][c];
            // z[j] -= z[c] L[j][c], so one instruction stream factorises and inverts (lanes 32-63 repeat).  L[j][c] is lane j's z[c] of
            // the FIRST row of lanes: row_newbcast:j spreads it over that row, row_bcast:15 hands it on to the second -- four 32-bit DPP
            // moves.  Measured per tile (scripts/dense_kbench.hip, 5.8K - 10K cycles depending on what the workers on the same SIMD do):
            // the same as round 3's v_readlane pair + FMA with a scalar operand -- about 40 cycles per (c, j) pair either way, 120 pairs
            // and 16 pivots per tile; broadcast LDS reads: 10.6K; rows and columns in the SAME lanes (two moves, two FMAs per pair):
            // 64 registers here, spills.  The chain stays the kernel's critical path.
            double z[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const double tv = dtile[k * 16 + idx];
                z[k] = is_w ? (k == idx ? 1.0 : 0.0) : tv;
            }
            double mypiv = 1.0;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const double piv = first_row_lane(z[c], c);
                notpd |= !(piv > 0.0);
                const double rs = rsqrt_fast(piv);
                const double l = z[c] * rs;        // rows: L[row][c]; columns: the final W[c][col]
                z[c] = l;
                mypiv = lane == c ? piv : mypiv;
#pragma unroll
                for (int j = c + 1; j < 16; ++j) {
                    double zj = fma(-l, first_row_lane(l, j), z[j]);      // L[j][c] comes from lane j (a row lane)
                    // evaluate NOW: left to itself the compiler sinks these updates to their use (a left-looking order) and keeps
                    // all 120 broadcast values alive in between
                    asm volatile("" : "+v"(zj));
                    z[j] = zj;
                }
            }
            if (lane < 16) {
                dg[kb * 16 + lane] = mypiv;
#pragma unroll
                for (int k = 0; k < 16; ++k) L[(kb * 16 + idx) + (int64_t)(kb * 16 + k) * ldL] = k <= idx ? z[k] : 0.0;
            } else if (lane < 32) {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    winv[kb & 1][idx * 16 + k] = z[k];              // W[k][col] at [k' = col][row' = k]
                    Dinv[kb * 256 + idx * 16 + k] = z[k];
                }
            }
#endif
            DK_STAMP(1);
            publish(&flag_w, kb + 1);
        }
        if (lane == 0 && notpd) bad = 1;
    } else {
        // ---------------------------------------------------------------- workers
        // Software pipeline per panel kb: U(kb) = the URGENT part of panel kb's rank-16 update (column kb+1, which the next
        // solve needs, and the diagonal tile after it), then the solve of column kb+1 as soon as the chain wave has published
        // its diagonal block, and only then R(kb) = the rest of panel kb's update, which overlaps the chain wave's next
        // factorisation. The panel buffer is a ring of three: panel kb is still read (R(kb)) after column kb+1 was written.
        auto solve = [&](int kb, int ln) __attribute__((always_inline)) {
            double* pb = pan[kb % 3];
            double wa[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) wa[ks] = winv[kb & 1][(ks * 4 + (ln >> 4)) * 16 + (ln & 15)];
#pragma unroll
            for (int s = 0; s < kCholSlots; ++s)
                if (TJ(s) == kb && TI(s) > kb) {
                    d4 x = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) x = mfma_f64(wa[ks], tile[s][ks], x);   // L_ik' = W T_ik'
                    tile[s] = x;
                    if (s + 1 < kCholSlots) {
                        if (TI(s) == kb + 1) {   // slot s + 1 is the diagonal tile (kb+1, kb+1): T -= L L' from registers, hand it over
                            d4 dt = tile[s + 1 < kCholSlots ? s + 1 : s];
#pragma unroll
                            for (int ks = 0; ks < 4; ++ks) dt = mfma_f64(-x[ks], x[ks], dt);
                            tile[s + 1 < kCholSlots ? s + 1 : s] = dt;
#pragma unroll
                            for (int r = 0; r < 4; ++r) dtile[r * 64 + ln] = dt[r];
                            publish(&flag_d, kb + 2);
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        pb[TI(s) * 256 + r * 64 + ln] = x[r];
                        L[(TI(s) * 16 + (ln & 15)) + (int64_t)(kb * 16 + (ln >> 4) + 4 * r) * ldL] = x[r];
                    }
                }
            // arrive at the worker barrier of this panel: the whole panel is in the LDS buffer before anyone applies it
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) __hip_atomic_fetch_add(&arrived, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        // rank-16 update with panel kb: urgent == tiles of column kb+1 and the diagonal tile (kb+2, kb+2); the diagonal tile
        // (kb+1, kb+1) got this update from its owner's registers during the solve
        auto update = [&](int kb, int ln, bool urgent) __attribute__((always_inline)) {
            const double* pb = pan[kb % 3];
            const int dnext = ((kb + 2) << 4) | (kb + 2);
#pragma unroll
            for (int s = 0; s < kCholSlots; ++s) {
                const bool urg = TJ(s) == kb + 1 || tt[s] == dnext;
                if (TJ(s) > kb && TI(s) > kb + 1 && urg == urgent) {
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        const int off = (ks * 4 + (ln >> 4)) * 16 + (ln & 15);
                        tile[s] = mfma_f64(-pb[TJ(s) * 256 + off], pb[TI(s) * 256 + off], tile[s]);
                    }
                }
            }
        };
        {
            int ln = lane;
            asm volatile("" : "+v"(ln));
            wait_ge(&flag_w, 1);
            solve(0, ln);
            wait_ge(&arrived, kCholWorkers);
        }
        for (int kb = 0; kb + 1 < nt; ++kb) {
            int ln = lane;   // opaque per panel: keeps the per-tile LDS addresses from being hoisted out of the panel loop
            asm volatile("" : "+v"(ln));
            DK_STAMP(2);
            update(kb, ln, true);
            wait_ge(&flag_w, kb + 2);
            solve(kb + 1, ln);
            DK_STAMP(3);
            update(kb, ln, false);
            wait_ge(&arrived, kCholWorkers * (kb + 2));
        }
    }
    __syncthreads();
    // log det S = sum log(pivot)
    double* redl = pan[0];
    redl[tid] = tid < n ? log(dg[tid]) : 0.0;
    __syncthreads();
    for (int st = 512; st > 0; st >>= 1) {
        if (tid < st) redl[tid] += redl[tid + st];
        __syncthreads();
    }
    if (tid == 0) {
        scal[0] = redl[0];
        scal[2] = bad ? 1.0 : 0.0;
        if (bad) scal[3] = 1.0;      // sticky: any factorisation of the call (the smoother's blocked d x d ones report through it)
    }
}
#undef TI
#undef TJ

// tile -> (wave, slot) table of dk_chol for an nt x nt tile grid: slots[kCholWorkers][kCholSlots], entries (i << 4) | j or -1. The pair
// {(i, i-1), (i, i)} goes to one wave in ADJACENT slots (see dk_chol); the other tiles go, in row order, to the wave that holds
// the fewest so far.
static void chol_slot_table(int nt, std::vector<int>& out) {
    std::vector<std::vector<int>> per(kCholWorkers);
    for (int i = 0; i < nt; ++i) {
        auto& v = per[i % kCholWorkers];
        if (i > 0) v.push_back((i << 4) | (i - 1));
        v.push_back((i << 4) | i);
    }
    for (int i = 2; i < nt; ++i)
        for (int j = 0; j + 1 < i; ++j) {
            int best = 0;
            for (int wv = 1; wv < kCholWorkers; ++wv)
                if (per[wv].size() < per[best].size()) best = wv;
            per[best].push_back((i << 4) | j);
        }
    out.assign(kCholWorkers * kCholSlots, -1);
    for (int wv = 0; wv < kCholWorkers; ++wv)
        for (size_t s = 0; s < per[wv].size() && s < (size_t)kCholSlots; ++s) out[wv * kCholSlots + s] = per[wv][s];
}

// ------------------------------------------------------------------------------------------------ B = L^-1 V (16 columns per workgroup)
// Right-looking block substitution with the D -> B-operand identity of the f64 MFMA layout: the accumulator of block row b
// (lane l, register r = row (l >> 4) + 4 r, column l & 15) IS the B operand of k-step r, so X_b = Dinv_b acc_b and the updates
// acc_b' -= L_b'b X_b chain through registers; only X_b crosses waves (LDS, double-buffered: one barrier per block row).
__global__ __launch_bounds__(256) void dk_trsm(const double* __restrict__ L, int64_t ldL, const double* __restrict__ Dinv,
                                               const double* __restrict__ V, int64_t ldV, int n, double* __restrict__ Bm, int64_t ldB) {
    __shared__ double xb[2][256];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = n / 16, c0 = blockIdx.x * 16;
    d4 acc[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int b = w + 4 * s;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            acc[s][r] = b < nb ? V[(b * 16 + (lane >> 4) + 4 * r) + (int64_t)(c0 + (lane & 15)) * ldV] : 0.0;
    }
    double dn[4];
    auto load_dinv = [&](int b) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) dn[ks] = b < nb ? Dinv[b * 256 + (ks * 4 + (lane >> 4)) * 16 + (lane & 15)] : 0.0;
    };
    load_dinv(w);
    // L fragments of block column b for the owned rows below b: loaded one stage ahead (they do not depend on X_b)
    d4 lfa[4], lfb[4];
    auto load_lf = [&](d4 (&lf)[4], int b) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int bb = w + 4 * s;
            const bool ok = bb > b && bb < nb && b < nb;
            const int bbc = ok ? bb : 0, bc = ok ? b : 0;     // unpredicated loads (clamped address), value selected
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const double v = L[(bbc * 16 + (lane & 15)) + (int64_t)(bc * 16 + ks * 4 + (lane >> 4)) * ldL];
                lf[s][ks] = ok ? v : 0.0;
            }
        }
    };
    auto stage = [&](int b, int s, bool owner, d4 (&lf)[4]) __attribute__((always_inline)) {
        if (owner) {
            d4 x = d4{0.0, 0.0, 0.0, 0.0};
            const d4 cur = acc[s];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) x = mfma_f64(dn[ks], cur[ks], x);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                xb[b & 1][r * 64 + lane] = x[r];
                Bm[(int64_t)(b * 16 + (lane >> 4) + 4 * r) * ldB + c0 + (lane & 15)] = x[r];
            }
            load_dinv(b + 4);
        }
        lds_barrier();
        double xr[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) xr[ks] = -xb[b & 1][ks * 64 + lane];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int bb = w + 4 * s;
            if (bb > b && bb < nb) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) acc[s] = mfma_f64(lf[s][ks], xr[ks], acc[s]);
            }
        }
    };
    load_lf(lfa, 0);
#pragma unroll
    for (int s = 0; s < 4; ++s) {          // owned block row of this round: b = 4 s + w (static s keeps acc[] in registers)
#pragma unroll
        for (int u = 0; u < 4; u += 2) {
            const int b = 4 * s + u;
            if (b < nb) {
                load_lf(lfb, b + 1);
                stage(b, s, u == w, lfa);
            }
            if (b + 1 < nb) {
                load_lf(lfa, b + 2);
                stage(b + 1, s, u + 1 == w, lfb);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ structured (sparse) A and H
// lgssm_components(::Separable, ...) builds A = I (x) A_t and H = I (x) H_t' as DENSE matrices (to_gauss_markov.jl:14-18) and the
// reference multiplies them as such. When a shared A or H has at most kSparseMaxNnz entries per row the engine keeps it in ELL
// form (col / val [nnz][rows], lanes run along rows) and forms A P, (A P) A' + Q, H Pp, V H' + R as row / column combinations:
// O(nnz d^2) streamed work instead of an O(d^3) contraction. Same values up to the order of the few remaining additions.
constexpr int kSparseMaxNnz = 8;

struct SpVec {   // optional vector product riding along dk_spl (blockIdx.y == gridDim.y - 1)
    int mode = 0;                    // 0 none; 1 out = add + Sp x; 2 residual out = y - h - Sp x (+ missing count)
    const double* x = nullptr;
    const double* add = nullptr;
    double* out = nullptr;
    const double* y = nullptr;
    const uint8_t* mask = nullptr;
    const double* hh = nullptr;
    int p = 0;
    double* scal = nullptr;
};

// out[i + k ldo] = sum_a val[a][i] in[col[a][i] + k ldi], k < ncols: every thread owns a row and walks KB columns
template <int KB> __global__ __launch_bounds__(256) void dk_spl(const int* __restrict__ col, const double* __restrict__ val, int nnz, int rows,
                                                                 const double* __restrict__ in, int64_t ldi, double* __restrict__ out,
                                                                 int64_t ldo, int ncols, SpVec v) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    int ci[kSparseMaxNnz];
    double cv[kSparseMaxNnz];
#pragma unroll
    for (int a = 0; a < kSparseMaxNnz; ++a) {
        const bool ok = a < nnz && i < rows;
        ci[a] = ok ? col[a * rows + i] : 0;
        cv[a] = ok ? val[a * rows + i] : 0.0;
    }
    if (blockIdx.y == gridDim.y - 1) {   // vector rider
        if (v.mode == 0) return;
        double s = 0.0;
#pragma unroll
        for (int a = 0; a < kSparseMaxNnz; ++a)
            if (a < nnz) s += cv[a] * v.x[ci[a]];
        if (i < rows) {
            if (v.mode == 2) {
                double o = 0.0;
                if (i < v.p) {
                    const bool miss = v.mask != nullptr && v.mask[i] != 0;
                    o = (miss ? 0.0 : v.y[i]) - v.hh[i] - s;
                }
                v.out[i] = o;
            } else {
                v.out[i] = s + (v.add ? v.add[i] : 0.0);
            }
        }
        if (v.mode == 2 && blockIdx.x == 0) {
            __shared__ int cnt;
            if (threadIdx.x == 0) cnt = 0;
            __syncthreads();
            int c = 0;
            if (v.mask)
                for (int q = threadIdx.x; q < v.p; q += 256) c += v.mask[q] != 0;
            if (c) atomicAdd(&cnt, c);
            __syncthreads();
            if (threadIdx.x == 0) v.scal[1] = (double)cnt;
        }
        return;
    }
    if (i >= rows) return;
    const int k0 = blockIdx.y * KB;
#pragma unroll
    for (int u = 0; u < KB; ++u) {
        const int k = k0 + u;
        if (k < ncols) {
            double s = 0.0;
#pragma unroll
            for (int a = 0; a < kSparseMaxNnz; ++a)
                if (a < nnz) s += cv[a] * in[ci[a] + (int64_t)k * ldi];
            out[i + (int64_t)k * ldo] = s;
        }
    }
}

// out[i + j ldo] = sum_b val[b][j] in[i + col[b][j] ldi] + E[i + j lde] (+ diag as in dk_gemm), i < M, j < N (N = rows of the
// sparse factor); a block owns 256 rows i and JB output columns, the sparse row of a column is wave-uniform
template <int JB> __global__ __launch_bounds__(256) void dk_spr(const int* __restrict__ col, const double* __restrict__ val, int nnz, int N,
                                                                 const double* __restrict__ in, int64_t ldi, double* __restrict__ out,
                                                                 int64_t ldo, int M, const double* __restrict__ E, int64_t lde,
                                                                 const double* __restrict__ diag, const uint8_t* __restrict__ dmask, int ndiag) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const int j0 = blockIdx.y * JB;
#pragma unroll
    for (int u = 0; u < JB; ++u) {
        const int j = j0 + u;
        if (j < N) {
            double s = 0.0;
#pragma unroll
            for (int b = 0; b < kSparseMaxNnz; ++b)
                if (b < nnz) s += val[b * N + j] * in[i + (int64_t)col[b * N + j] * ldi];
            if (E) s += E[i + (int64_t)j * lde];
            if (diag && i == j) s += i < ndiag ? ((dmask && dmask[i]) ? kLargeVar : diag[i]) : 1.0;
            out[i + (int64_t)j * ldo] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------------ prior marginals of one step
// mean_i = (H mp + h)_i (rider of the V = H Pp launch writes -r = H mp + h - 0 ... see host code), var_i = sum_k V[i][k] H[i][k] + R_i
__global__ void __launch_bounds__(256) dk_marg_diag(const double* __restrict__ V, const double* __restrict__ Hk, int Pq, int Dp,
                                                    const double* __restrict__ R, const double* __restrict__ res, int p,
                                                    double* __restrict__ mean_out, double* __restrict__ var_out) {
    // 16 rows x 16 K-slices per workgroup; the slices are summed in a fixed order
    __shared__ double part[16][17];
    const int r = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + r;            // < Pq (the grid covers ceil(p / 16) <= Pq / 16 row tiles)
    double s = 0.0;
    for (int k = sl; k < Dp; k += 16) s += V[i + (int64_t)k * Pq] * Hk[i + (int64_t)k * Pq];
    part[sl][r] = s;
    __syncthreads();
    if (sl == 0 && i < p) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += part[q][r];
        var_out[i] = t + R[i];
        mean_out[i] = -res[i];   // res = 0 - h - H mp
    }
}

__global__ void dk_pack(const double* __restrict__ src, int64_t rs, int64_t cs, int rows, int cols, double* __restrict__ dst, int64_t ldd,
                        int prow, int pcol, double padval_diag) {
    // dst[i + j ldd] = (i < rows && j < cols) ? src[i rs + j cs] : (i == j ? padval_diag : 0), for i < prow, j < pcol
    const int64_t n = (int64_t)prow * pcol;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(e % prow), j = (int)(e / prow);
        dst[i + (int64_t)j * ldd] = (i < rows && j < cols) ? src[(int64_t)i * rs + (int64_t)j * cs] : ((i == j) ? padval_diag : 0.0);
    }
}

// ================================================================================================ host side
struct Buf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        hipError_t e = hipMalloc(&p, bytes + kOperandSlack * sizeof(double));   // see dk_gemm: unpredicated edge loads
        if (e == hipSuccess) cap = bytes;
        if (e == hipSuccess) {       // TGP_POISON=1: NaN-fill fresh allocations (debugging aid, see tgp_api.hip)
            static const bool poison = [] { const char* v = std::getenv("TGP_POISON"); return v != nullptr && v[0] == '1'; }();
            if (poison) {
                e = hipMemset(p, 0xFF, bytes + kOperandSlack * sizeof(double));
                (void)hipDeviceSynchronize();
            }
        }
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    double* d() const { return static_cast<double*>(p); }
};

struct Prof {
    std::string name;
    double ms = 0.0;
    int64_t calls = 0;
};

struct Engine {
    int device = 0;
    std::string err;
    bool have_model = false;
    int64_t T = 0;
    int d = 0, p = 0, ordering = 0, Dp = 0, Pq = 0;
    int64_t ldB = 0;
    // packed model (padded); per-step arrays keep a stride
    Buf bA, bQ, bH, ba, bh, bR, bx0;
    int64_t sA = 0, sQ = 0, sH = 0, sa = 0, sh = 0, sR = 0;
    // state and work buffers
    Buf bm, bmp, bP, bPp, bT1, bV, bS, bL, bDinv, bB, bscal, bslots;
    // RTS smoother (posterior_marginals): stored filtering states, the blocked d x d Cholesky factor, work matrices
    Buf bPstore, bmstore, bLd, bDinvd, bW0, bW1, bW2, bW3, bslots_blk, bslots_tail, bzero, bPbound, bmbound;
    int fused_opt = 1;           // mid-sized states (Dp <= 64, p <= 16): the persistent single-kernel passes of tgp_dense_fused.hpp
    Buf bfin;                    // state handed from one launch of a persistent pass to the next
    int64_t segment_opt = 0;     // smoother segment length (0 = automatic); tests force small segments
    // ELL form of a shared A / H with few entries per row (0 == dense)
    int structure_opt = 1;
    int nnzA = 0, nnzH = 0;
    Buf bAcol, bAval, bHcol, bHval;
    int profile = 0;
    std::vector<Prof> prof;
    struct Pending {
        int idx;
        hipEvent_t a, b;
    };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> pool;
    bool attrs_set = false;
    int fail(int code, const std::string& m) {
        err = m;
        return code;
    }
};

#define DCHK(expr)                                                                                   \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) return e->fail(TGP_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

Engine* create(int device) {
    Engine* e = new Engine();
    e->device = device;
    return e;
}
void destroy(Engine* e) {
    if (!e) return;
    for (Buf* b : {&e->bA, &e->bQ, &e->bH, &e->ba, &e->bh, &e->bR, &e->bx0, &e->bm, &e->bmp, &e->bP, &e->bPp, &e->bT1, &e->bV, &e->bS,
                   &e->bL, &e->bDinv, &e->bB, &e->bscal, &e->bslots, &e->bAcol, &e->bAval, &e->bHcol, &e->bHval, &e->bPstore, &e->bmstore, &e->bLd, &e->bDinvd, &e->bW0, &e->bW1, &e->bW2, &e->bW3,
                   &e->bslots_blk, &e->bslots_tail, &e->bzero, &e->bPbound, &e->bmbound, &e->bfin})
        b->release();
    for (auto& pe : e->pending) {
        (void)hipEventDestroy(pe.a);
        (void)hipEventDestroy(pe.b);
    }
    for (auto ev : e->pool) (void)hipEventDestroy(ev);
    delete e;
}
const std::string& last_error(const Engine* e) { return e->err; }
void set_profile(Engine* e, int on) { e->profile = on; }
void set_structure(Engine* e, int on) { e->structure_opt = on; }
void set_segment(Engine* e, int64_t steps) { e->segment_opt = steps; }
void set_fused(Engine* e, int on) { e->fused_opt = on; }
int fused(const Engine* e) { return e->fused_opt && e->Dp <= 64 && e->p <= 16; }
int structure(const Engine* e) { return (e->nnzA ? 1 : 0) | (e->nnzH ? 2 : 0) | ((e->fused_opt && e->Dp <= 64 && e->p <= 16) ? 4 : 0); }
static void resolve_pending(Engine* e);
int profile_count(Engine* e) {
    resolve_pending(e);   // callers query after the call's closing stream synchronisation
    return (int)e->prof.size();
}
KernelTime profile_get(const Engine* e, int idx) { return KernelTime{e->prof[idx].name.c_str(), e->prof[idx].ms, e->prof[idx].calls}; }
void profile_reset(Engine* e) {
    resolve_pending(e);
    e->prof.clear();
}

namespace {

struct Scope {   // hipEvent bracket of one launch (profile mode only)
    Engine* e;
    hipStream_t st;
    int idx = -1;
    hipEvent_t a = nullptr, b = nullptr;
    Scope(Engine* e_, hipStream_t st_, const char* name, bool on) : e(e_), st(st_) {
        if (!on) return;
        for (size_t i = 0; i < e->prof.size(); ++i)
            if (e->prof[i].name == name) idx = (int)i;
        if (idx < 0) {
            e->prof.push_back(Prof{name, 0.0, 0});
            idx = (int)e->prof.size() - 1;
        }
        auto get = [&]() {
            hipEvent_t ev = nullptr;
            if (!e->pool.empty()) {
                ev = e->pool.back();
                e->pool.pop_back();
            } else {
                (void)hipEventCreate(&ev);
            }
            return ev;
        };
        a = get();
        b = get();
        (void)hipEventRecord(a, st);
    }
    ~Scope() {
        if (idx < 0) return;
        (void)hipEventRecord(b, st);
        e->pending.push_back(Engine::Pending{idx, a, b});
    }
};

}  // namespace
static void resolve_pending(Engine* e) {
    for (auto& pe : e->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, pe.a, pe.b) == hipSuccess) {
            e->prof[pe.idx].ms += ms;
            e->prof[pe.idx].calls += 1;
        }
        e->pool.push_back(pe.a);
        e->pool.push_back(pe.b);
    }
    e->pending.clear();
}
namespace {
void resolve(Engine* e) { resolve_pending(e); }

template <int TM, int TN> void launch_gemm_t(GemmArgs& g, hipStream_t st) {
    using Cfg = GemmCfg<TM, TN>;
    const int ntm = (g.M + Cfg::BM - 1) / Cfg::BM, ntn = (g.N + Cfg::BN - 1) / Cfg::BN;
    const int per = (ntm * ntn + 7) / 8;
    g.nblk_tiles = per * 8;
    const int nrider = g.v.mode ? (g.v.n + 15) / 16 : 0;
    hipLaunchKernelGGL((dk_gemm<TM, TN>), dim3(g.nblk_tiles + nrider), dim3(512), Cfg::LDS_BYTES, st, g);
}

// tile shape by problem size: 48 x 48 blocks once they fill the 256 CUs, 16 x 48 for the p x d products, else single tiles
void launch_gemm(GemmArgs& g, hipStream_t st) {
    const int t33 = ((g.M + 47) / 48) * ((g.N + 47) / 48);
    const int t13 = ((g.M + 15) / 16) * ((g.N + 47) / 48);
    if (t33 >= 160) launch_gemm_t<3, 3>(g, st);
    else if (t13 >= 160) launch_gemm_t<1, 3>(g, st);
    else launch_gemm_t<1, 1>(g, st);
}

int set_attrs(Engine* e) {
    if (e->attrs_set) return TGP_OK;
    DCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dk_gemm<3, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GemmCfg<3, 3>::LDS_BYTES));
    DCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dk_gemm<1, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GemmCfg<1, 3>::LDS_BYTES));
    DCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dk_gemm<1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GemmCfg<1, 1>::LDS_BYTES));
    DCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dk_fused_filter<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FusedCfg<32>::LDS_BYTES));
    DCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dk_fused_filter<48>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FusedCfg<48>::LDS_BYTES));
    DCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dk_fused_filter<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FusedCfg<64>::LDS_BYTES));
    DCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dk_fused_rand<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FusedRandCfg<32>::LDS_BYTES));
    DCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dk_fused_rand<48>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FusedRandCfg<48>::LDS_BYTES));
    DCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dk_fused_rand<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FusedRandCfg<64>::LDS_BYTES));
    DCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dk_fused_smooth<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FusedSmoothCfg<32>::LDS_BYTES));
    DCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dk_fused_smooth<48>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FusedSmoothCfg<48>::LDS_BYTES));
    DCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dk_fused_smooth<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FusedSmoothCfg<64>::LDS_BYTES));
    e->attrs_set = true;
    return TGP_OK;
}

inline int rup16(int x) { return (x + 15) / 16 * 16; }

}  // namespace

int model_set(Engine* e, const ModelDesc& m, hipStream_t st) {
    e->have_model = false;
    DCHK(hipSetDevice(e->device));
    if (int rc = set_attrs(e)) return rc;
    if (m.p > kCholMaxTiles * 16) return e->fail(TGP_EUNSUPPORTED, "dense path: observation dimension p must be <= 256");
    if (m.p > m.d + 16 * 64) return e->fail(TGP_EUNSUPPORTED, "dense path: p too large");
    e->T = m.T;
    e->d = m.d;
    e->p = m.p;
    e->ordering = m.ordering;
    const int Dp = rup16(m.d), Pq = rup16(m.p);
    e->Dp = Dp;
    e->Pq = Pq;
    e->ldB = Dp + 16;
    auto steps = [&](int64_t s) { return s ? m.T : (int64_t)1; };
    const size_t DD = (size_t)Dp * Dp, PD = (size_t)Pq * Dp;
    DCHK(e->bA.ensure(steps(m.sA) * DD * 8));
    DCHK(e->bQ.ensure(steps(m.sQ) * DD * 8));
    DCHK(e->bH.ensure(steps(m.sH) * PD * 8));
    DCHK(e->ba.ensure(steps(m.sa) * (size_t)Dp * 8));
    DCHK(e->bh.ensure(steps(m.sh) * (size_t)Pq * 8));
    DCHK(e->bR.ensure(steps(m.sR) * (size_t)Pq * 8));
    e->sA = m.sA ? (int64_t)DD : 0;
    e->sQ = m.sQ ? (int64_t)DD : 0;
    e->sH = m.sH ? (int64_t)PD : 0;
    e->sa = m.sa ? Dp : 0;
    e->sh = m.sh ? Pq : 0;
    e->sR = m.sR ? Pq : 0;
    auto pack = [&](const double* src, int64_t sstride, int64_t nsteps, int64_t rs, int64_t cs, int rows, int cols, double* dst, int64_t dstride,
                    int prow, int pcol, double padv) {
        for (int64_t t = 0; t < nsteps; ++t)
            hipLaunchKernelGGL(dk_pack, dim3(64), dim3(256), 0, st, src + t * sstride, rs, cs, rows, cols, dst + t * dstride, (int64_t)prow, prow, pcol, padv);
    };
    // A, Q: column-major d x d -> column-major Dp x Dp
    pack(m.A, m.sA, steps(m.sA), 1, m.d, m.d, m.d, e->bA.d(), DD, Dp, Dp, 0.0);
    pack(m.Q, m.sQ, steps(m.sQ), 1, m.d, m.d, m.d, e->bQ.d(), DD, Dp, Dp, 0.0);
    // H [p][d] row-major (H[i][k] at i d + k) -> Hk[k Pq + i]: "rows" = i (stride d), "cols" = k (stride 1)
    pack(m.H, m.sH, steps(m.sH), m.d, 1, m.p, m.d, e->bH.d(), PD, Pq, Dp, 0.0);
    pack(m.a, m.sa, steps(m.sa), 1, 0, m.d, 1, e->ba.d(), Dp, Dp, 1, 0.0);
    pack(m.h, m.sh, steps(m.sh), 1, 0, m.p, 1, e->bh.d(), Pq, Pq, 1, 0.0);
    // R: pad with ones (column vector: the "diagonal" pad rule of dk_pack only hits i == j == 0, so pad explicitly below)
    {
        std::vector<double> ones((size_t)Pq, 1.0);
        for (int64_t t = 0; t < steps(m.sR); ++t) {
            DCHK(hipMemcpyAsync(e->bR.d() + t * Pq, ones.data(), (size_t)Pq * 8, hipMemcpyHostToDevice, st));
            DCHK(hipMemcpyAsync(e->bR.d() + t * Pq, m.R + t * m.sR, (size_t)m.p * 8, hipMemcpyDeviceToDevice, st));
        }
        DCHK(hipStreamSynchronize(st));
    }
    // structure of a shared A / H (see dk_spl): ELL form when no row has more than kSparseMaxNnz entries
    e->nnzA = e->nnzH = 0;
    if (e->structure_opt) {
        auto build = [&](const double* dev, bool shared, int rows, int cols, int64_t rs, int64_t cs, int prows, Buf& bcol, Buf& bval, int& nnz_out) -> int {
            if (!shared) return TGP_OK;
            std::vector<double> host((size_t)rows * cols);
            DCHK(hipMemcpy(host.data(), dev, host.size() * 8, hipMemcpyDeviceToHost));
            int mx = 0;
            for (int i = 0; i < rows && mx <= kSparseMaxNnz; ++i) {
                int c = 0;
                for (int k = 0; k < cols; ++k) c += host[(size_t)i * rs + (size_t)k * cs] != 0.0;
                mx = c > mx ? c : mx;
            }
            if (mx == 0 || mx > kSparseMaxNnz) return TGP_OK;
            std::vector<int> col((size_t)mx * prows, 0);
            std::vector<double> val((size_t)mx * prows, 0.0);
            for (int i = 0; i < rows; ++i) {
                int c = 0;
                for (int k = 0; k < cols; ++k) {
                    const double v = host[(size_t)i * rs + (size_t)k * cs];
                    if (v != 0.0) {
                        col[(size_t)c * prows + i] = k;
                        val[(size_t)c * prows + i] = v;
                        ++c;
                    }
                }
            }
            DCHK(bcol.ensure(col.size() * sizeof(int)));
            DCHK(bval.ensure(val.size() * 8));
            DCHK(hipMemcpy(bcol.p, col.data(), col.size() * sizeof(int), hipMemcpyHostToDevice));
            DCHK(hipMemcpy(bval.p, val.data(), val.size() * 8, hipMemcpyHostToDevice));
            nnz_out = mx;
            return TGP_OK;
        };
        DCHK(hipStreamSynchronize(st));
        if (int rc = build(m.A, m.sA == 0, m.d, m.d, 1, m.d, Dp, e->bAcol, e->bAval, e->nnzA)) return rc;      // A[i][k] at i + k d
        if (int rc = build(m.H, m.sH == 0, m.p, m.d, m.d, 1, Pq, e->bHcol, e->bHval, e->nnzH)) return rc;      // H[i][k] at i d + k
    }
    DCHK(e->bx0.ensure((DD + Dp) * 8));
    DCHK(e->bm.ensure((size_t)Dp * 8));
    DCHK(e->bmp.ensure((size_t)Dp * 8));
    DCHK(e->bP.ensure(DD * 8));
    DCHK(e->bPp.ensure(DD * 8));
    DCHK(e->bT1.ensure(DD * 8));
    DCHK(e->bV.ensure((size_t)Pq * (Dp + 16) * 8));
    DCHK(e->bS.ensure((size_t)Pq * Pq * 8));
    DCHK(e->bL.ensure((size_t)Pq * Pq * 8));
    DCHK(e->bDinv.ensure((size_t)(Pq / 16) * 256 * 8));
    DCHK(e->bB.ensure((size_t)Pq * e->ldB * 8));
    DCHK(e->bscal.ensure(8 * 8));
    {
        std::vector<int> tab;
        chol_slot_table(Pq / 16, tab);
        DCHK(e->bslots.ensure(tab.size() * sizeof(int)));
        DCHK(hipMemcpyAsync(e->bslots.p, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice, st));
        DCHK(hipStreamSynchronize(st));
    }
    DCHK(hipMemsetAsync(e->bV.p, 0, (size_t)Pq * (Dp + 16) * 8, st));
    DCHK(hipMemsetAsync(e->bL.p, 0, (size_t)Pq * Pq * 8, st));
    DCHK(hipMemsetAsync(e->bscal.p, 0, 64, st));
    if (int rc = set_x0(e, m.x0m, m.x0P, st)) return rc;
    e->have_model = true;
    return TGP_OK;
}

int set_x0(Engine* e, const double* x0m, const double* x0P, hipStream_t st) {
    const int Dp = e->Dp, d = e->d;
    std::vector<double> buf((size_t)Dp * Dp + Dp, 0.0);
    for (int j = 0; j < d; ++j)
        for (int i = 0; i < d; ++i) buf[i + (size_t)j * Dp] = x0P[i + (size_t)j * d];
    for (int i = 0; i < d; ++i) buf[(size_t)Dp * Dp + i] = x0m[i];
    DCHK(hipMemcpyAsync(e->bx0.p, buf.data(), buf.size() * 8, hipMemcpyHostToDevice, st));
    DCHK(hipStreamSynchronize(st));
    return TGP_OK;
}

namespace {

struct StepPtrs {
    const double *A, *Q, *H, *a, *h, *R;
};
StepPtrs step_ptrs(const Engine* e, int64_t t) {
    return StepPtrs{e->bA.d() + t * e->sA, e->bQ.d() + t * e->sQ, e->bH.d() + t * e->sH, e->ba.d() + t * e->sa, e->bh.d() + t * e->sh,
                    e->bR.d() + t * e->sR};
}

// predict (lgc.jl:46-52): mp = A m + a, Pp = A P A' + Q
void enqueue_predict(Engine* e, const StepPtrs& s, hipStream_t st, bool prof) {
    const int Dp = e->Dp;
    if (e->nnzA) {
        {
            SpVec v;
            v.mode = 1; v.x = e->bm.d(); v.add = s.a; v.out = e->bmp.d();
            Scope sc(e, st, "dk_spl<A P>", prof);
            hipLaunchKernelGGL(dk_spl<2>, dim3((Dp + 255) / 256, (Dp + 1) / 2 + 1), dim3(256), 0, st, static_cast<const int*>(e->bAcol.p), e->bAval.d(),
                               e->nnzA, Dp, e->bP.d(), (int64_t)Dp, e->bT1.d(), (int64_t)Dp, Dp, v);
        }
        {
            Scope sc(e, st, "dk_spr<(A P) A' + Q>", prof);
            hipLaunchKernelGGL(dk_spr<1>, dim3((Dp + 255) / 256, Dp), dim3(256), 0, st, static_cast<const int*>(e->bAcol.p), e->bAval.d(), e->nnzA,
                               Dp, e->bT1.d(), (int64_t)Dp, e->bPp.d(), (int64_t)Dp, Dp, s.Q, (int64_t)Dp, (const double*)nullptr,
                               (const uint8_t*)nullptr, 0);
        }
        return;
    }
    {
        GemmArgs g;
        g.A = s.A; g.lda = Dp;
        g.B = e->bP.d(); g.ldb = Dp;
        g.C = e->bT1.d(); g.ldc = Dp;
        g.M = g.N = g.K = Dp;
        g.v.mode = 1;
        g.v.Mx = s.A; g.v.ld = Dp; g.v.n = Dp; g.v.K = Dp;
        g.v.x = e->bm.d(); g.v.xs = 1; g.v.add = s.a; g.v.out = e->bmp.d();
        Scope sc(e, st, "dk_gemm<A P>", prof);
        launch_gemm(g, st);
    }
    {
        GemmArgs g;
        g.A = e->bT1.d(); g.lda = Dp;
        g.B = s.A; g.ldb = Dp;           // Bop[k][j] = A[j][k] = Ak[j + k Dp]
        g.C = e->bPp.d(); g.ldc = Dp;
        g.E = s.Q; g.lde = Dp;
        g.M = g.N = g.K = Dp;
        Scope sc(e, st, "dk_gemm<(A P) A' + Q>", prof);
        launch_gemm(g, st);
    }
}

// posterior_and_lml (lgc.jl:129-151) on (mp, Pp) -> (m, P); lml_t accumulated into result8
void enqueue_update(Engine* e, const StepPtrs& s, int64_t t, const double* y, const uint8_t* mask, double* m_out, double* P_out, double* result8,
                    hipStream_t st, bool prof, int out_n = 0) {
    if (out_n == 0) out_n = e->d;       // leading dimension / size of the per-step output blocks (d for the API, Dp for the smoother's store)
    const int Dp = e->Dp, Pq = e->Pq;
    const double* yt = y + t * e->p;
    const uint8_t* mt = mask ? mask + t * e->p : nullptr;
    if (e->nnzH) {
        {   // V = H Pp; rider: r = y - h - H mp -> V[:, Dp]
            SpVec v;
            v.mode = 2; v.x = e->bmp.d(); v.out = e->bV.d() + (size_t)Dp * Pq;
            v.y = yt; v.mask = mt; v.hh = s.h; v.p = e->p; v.scal = e->bscal.d();
            Scope sc(e, st, "dk_spl<H Pp>", prof);
            hipLaunchKernelGGL(dk_spl<2>, dim3((Pq + 255) / 256, (Dp + 1) / 2 + 1), dim3(256), 0, st, static_cast<const int*>(e->bHcol.p), e->bHval.d(),
                               e->nnzH, Pq, e->bPp.d(), (int64_t)Dp, e->bV.d(), (int64_t)Pq, Dp, v);
        }
        {   // S = V H' + R
            Scope sc(e, st, "dk_spr<V H' + R>", prof);
            hipLaunchKernelGGL(dk_spr<1>, dim3((Pq + 255) / 256, Pq), dim3(256), 0, st, static_cast<const int*>(e->bHcol.p), e->bHval.d(), e->nnzH,
                               Pq, e->bV.d(), (int64_t)Pq, e->bS.d(), (int64_t)Pq, Pq, (const double*)nullptr, (int64_t)0, s.R, mt, e->p);
        }
    } else {
    {   // V = H Pp; rider: r = y - h - H mp -> V[:, Dp]
        GemmArgs g;
        g.A = s.H; g.lda = Pq;
        g.B = e->bPp.d(); g.ldb = Dp;
        g.C = e->bV.d(); g.ldc = Pq;
        g.M = Pq; g.N = Dp; g.K = Dp;
        g.v.mode = 2;
        g.v.Mx = s.H; g.v.ld = Pq; g.v.n = Pq; g.v.K = Dp;
        g.v.x = e->bmp.d(); g.v.xs = 1;
        g.v.out = e->bV.d() + (size_t)Dp * Pq;
        g.v.y = yt; g.v.mask = mt; g.v.hh = s.h; g.v.p = e->p;
        g.v.scal = e->bscal.d();
        Scope sc(e, st, "dk_gemm<H Pp>", prof);
        launch_gemm(g, st);
    }
    {   // S = V H' + R
        GemmArgs g;
        g.A = e->bV.d(); g.lda = Pq;
        g.B = s.H; g.ldb = Pq;
        g.C = e->bS.d(); g.ldc = Pq;
        g.diag = s.R; g.dmask = mt; g.ndiag = e->p;
        g.M = g.N = Pq; g.K = Dp;
        Scope sc(e, st, "dk_gemm<V H' + R>", prof);
        launch_gemm(g, st);
    }
    }
    {
        Scope sc(e, st, "dk_chol", prof);
        hipLaunchKernelGGL(dk_chol, dim3(1), dim3(1024), 0, st, e->bS.d(), (int64_t)Pq, Pq, 0.0, static_cast<const int*>(e->bslots.p), e->bL.d(), (int64_t)Pq,
                           e->bDinv.d(), e->bscal.d()
#ifdef DK_TRACE
                           , (long long*)nullptr
#endif
        );
    }
    {
        Scope sc(e, st, "dk_trsm", prof);
        hipLaunchKernelGGL(dk_trsm, dim3((Dp + 16) / 16), dim3(256), 0, st, e->bL.d(), (int64_t)Pq, e->bDinv.d(), e->bV.d(), (int64_t)Pq, Pq, e->bB.d(), e->ldB);
    }
    {   // P = Pp - B'B; rider: m = mp + B' alpha, lml_t
        GemmArgs g;
        g.A = e->bB.d(); g.lda = e->ldB;
        g.B = e->bB.d(); g.ldb = e->ldB;
        g.C = e->bP.d(); g.ldc = Dp;
        g.E = e->bPp.d(); g.lde = Dp;
        g.sign = -1.0;
        g.M = g.N = Dp; g.K = Pq;
        if (P_out) {
            g.C2 = P_out + t * (int64_t)out_n * out_n; g.ldc2 = out_n; g.M2 = g.N2 = out_n;
        }
        g.v.mode = 3;
        g.v.Mx = e->bB.d(); g.v.ld = e->ldB; g.v.n = Dp; g.v.K = Pq;
        g.v.x = e->bB.d() + Dp; g.v.xs = e->ldB;
        g.v.add = e->bmp.d(); g.v.out = e->bm.d();
        if (m_out) {
            g.v.out2 = m_out + t * out_n; g.v.n2 = out_n;
        }
        g.v.p = e->p; g.v.scal = e->bscal.d(); g.v.stats = result8; g.v.tstep = t;
        Scope sc(e, st, "dk_gemm<Pp - B'B>", prof);
        launch_gemm(g, st);
    }
}

}  // namespace

namespace {
constexpr int64_t kFusedStepsPerLaunch = 1 << 21;      // ~1 s of a persistent pass per launch at most

FusedArgs fused_args(const Engine* e, const double* y, const uint8_t* mask, double* result8) {
    FusedArgs g;
    g.T = e->T; g.d = e->d; g.p = e->p; g.Pq = e->Pq; g.ordering = e->ordering;
    g.A = e->bA.d(); g.Q = e->bQ.d(); g.H = e->bH.d(); g.a = e->ba.d(); g.h = e->bh.d(); g.R = e->bR.d();
    g.sA = e->sA; g.sQ = e->sQ; g.sH = e->sH; g.sa = e->sa; g.sh = e->sh; g.sR = e->sR;
    g.y = y; g.mask = mask; g.result8 = result8;
    return g;
}

// the filter pass as persistent launches of dk_fused_filter (the state travels through e->bfin between launches)
int fused_filter(Engine* e, const double* y, const uint8_t* mask, double* m_out, double* P_out, double* result8, hipStream_t st, double* aux_out = nullptr) {
    const size_t nst = ((size_t)e->Dp * e->Dp + e->Dp) * 8;
    DCHK(e->bfin.ensure(nst));
    FusedArgs g = fused_args(e, y, mask, result8);
    g.m_out = m_out; g.P_out = P_out; g.aux_out = aux_out;
    g.xfin = e->bfin.d();
    for (int64_t s0 = 0; s0 < e->T; s0 += kFusedStepsPerLaunch) {
        g.step0 = s0;
        g.step1 = std::min(e->T, s0 + kFusedStepsPerLaunch);
        g.x0 = s0 == 0 ? e->bx0.d() : e->bfin.d();
        Scope sc(e, st, "dk_fused_filter", e->profile != 0);
        if (e->Dp == 32) hipLaunchKernelGGL(dk_fused_filter<32>, dim3(1), dim3(256), FusedCfg<32>::LDS_BYTES, st, g);
        else if (e->Dp == 48) hipLaunchKernelGGL(dk_fused_filter<48>, dim3(1), dim3(256), FusedCfg<48>::LDS_BYTES, st, g);
        else hipLaunchKernelGGL(dk_fused_filter<64>, dim3(1), dim3(256), FusedCfg<64>::LDS_BYTES, st, g);
    }
    DCHK(hipGetLastError());
    return TGP_OK;
}
// posterior marginals of a mid-sized Forward model: the persistent filter keeps (m_t, P_t) and the per-update records, the
// persistent Bryson-Frazier pass walks back (tgp_dense_fused.hpp). Returns TGP_EUNSUPPORTED (quietly, no message) when the
// stored states do not fit: the caller then runs the segmented chain.
int fused_posterior_marginals(Engine* e, const double* y, const uint8_t* mask, const double* Rnew, int64_t sRn, double* mean_out, double* var_out,
                              double* result8, hipStream_t st) {
    const int d = e->d, p = e->p, Dp = e->Dp;
    const size_t nP = (size_t)e->T * d * d * 8, nm = (size_t)e->T * d * 8, nx = (size_t)e->T * p * (d + 2) * 8;
    size_t free_b = 0, total_b = 0;
    DCHK(hipMemGetInfo(&free_b, &total_b));
    if ((double)(nP + nm + nx) > 0.8 * ((double)free_b + (double)e->bPstore.cap + (double)e->bmstore.cap + (double)e->bPbound.cap)) return TGP_EUNSUPPORTED;
    DCHK(e->bPstore.ensure(nP));
    DCHK(e->bmstore.ensure(nm));
    DCHK(e->bPbound.ensure(nx));            // (the boundary buffer of the segmented smoother doubles as the record store)
    DCHK(e->bfin.ensure(((size_t)Dp * Dp + Dp) * 8));
    const int rcf = fused_filter(e, y, mask, e->bmstore.d(), e->bPstore.d(), result8, st, e->bPbound.d());
    if (rcf != TGP_OK) return rcf;
    Buf adj;
    DCHK(adj.ensure(((size_t)Dp * Dp + Dp) * 8));
    FusedSmoothArgs g;
    g.T = e->T; g.d = d; g.p = p; g.Pq = e->Pq;
    g.A = e->bA.d(); g.H = e->bH.d(); g.h = e->bh.d();
    g.sA = e->sA; g.sH = e->sH; g.sh = e->sh;
    g.m_f = e->bmstore.d(); g.P_f = e->bPstore.d(); g.aux = e->bPbound.d();
    g.Rnew = Rnew; g.sRn = sRn;
    g.adj = adj.d();
    g.mean_out = mean_out; g.var_out = var_out;
    for (int64_t s1 = e->T; s1 > 0; s1 -= kFusedStepsPerLaunch) {
        g.step1 = s1;
        g.step0 = std::max<int64_t>(0, s1 - kFusedStepsPerLaunch);
        g.first = s1 == e->T;
        Scope sc(e, st, "dk_fused_smooth", e->profile != 0);
        if (Dp == 32) hipLaunchKernelGGL(dk_fused_smooth<32>, dim3(1), dim3(256), FusedSmoothCfg<32>::LDS_BYTES, st, g);
        else if (Dp == 48) hipLaunchKernelGGL(dk_fused_smooth<48>, dim3(1), dim3(256), FusedSmoothCfg<48>::LDS_BYTES, st, g);
        else hipLaunchKernelGGL(dk_fused_smooth<64>, dim3(1), dim3(256), FusedSmoothCfg<64>::LDS_BYTES, st, g);
    }
    DCHK(hipStreamSynchronize(st));
    resolve(e);
    adj.release();
    return TGP_OK;
}
}  // namespace

int filter(Engine* e, const double* y, const uint8_t* mask, double* m_out, double* P_out, double* result8, hipStream_t st) {
    if (!e->have_model) return e->fail(TGP_EINVAL, "no model");
    DCHK(hipSetDevice(e->device));
    if (fused(e)) return fused_filter(e, y, mask, m_out, P_out, result8, st);
    const int Dp = e->Dp;
    const size_t DD = (size_t)Dp * Dp;
    DCHK(hipMemcpyAsync(e->bP.p, e->bx0.p, DD * 8, hipMemcpyDeviceToDevice, st));
    DCHK(hipMemcpyAsync(e->bm.p, e->bx0.d() + DD, (size_t)Dp * 8, hipMemcpyDeviceToDevice, st));
    for (int64_t step = 0; step < e->T; ++step) {
        const int64_t t = e->ordering == 0 ? step : e->T - 1 - step;
        const StepPtrs s = step_ptrs(e, t);
        // profile mode: events on every 16th step (an event pair per launch slows the host enqueue)
        const bool prof = e->profile && (step % 16 == 8 || e->T < 64);
        if (e->ordering == 0) {
            enqueue_predict(e, s, st, prof);
            enqueue_update(e, s, t, y, mask, m_out, P_out, result8, st, prof);
        } else {
            // Reverse (lgssm.jl:161-165,183-187): update from the carried state, then predict with the same step's transition.
            // The carried state lives in (m, P); the update reads (mp, Pp): swap roles by copying (d^2 doubles, L2-resident).
            DCHK(hipMemcpyAsync(e->bPp.p, e->bP.p, DD * 8, hipMemcpyDeviceToDevice, st));
            DCHK(hipMemcpyAsync(e->bmp.p, e->bm.p, (size_t)Dp * 8, hipMemcpyDeviceToDevice, st));
            enqueue_update(e, s, t, y, mask, m_out, P_out, result8, st, prof);
            enqueue_predict(e, s, st, prof);
            DCHK(hipMemcpyAsync(e->bP.p, e->bPp.p, DD * 8, hipMemcpyDeviceToDevice, st));
            DCHK(hipMemcpyAsync(e->bm.p, e->bmp.p, (size_t)Dp * 8, hipMemcpyDeviceToDevice, st));
        }
        if ((step & 1023) == 1023) {
            DCHK(hipStreamSynchronize(st));   // bound the host's run-ahead (and the event pool in profile mode)
            resolve(e);
        }
    }
    DCHK(hipGetLastError());
    return TGP_OK;
}

int marginals(Engine* e, double* mean_out, double* var_out, double* result8, hipStream_t st) {
    if (!e->have_model) return e->fail(TGP_EINVAL, "no model");
    DCHK(hipSetDevice(e->device));
    if (fused(e)) {       // mid-sized state: the persistent pass in its marginals mode (predict + emission marginal per step)
        DCHK(e->bfin.ensure(((size_t)e->Dp * e->Dp + e->Dp) * 8));
        FusedArgs g = fused_args(e, nullptr, nullptr, result8);
        g.marg_mean = mean_out; g.marg_var = var_out;
        g.xfin = e->bfin.d();
        for (int64_t s0 = 0; s0 < e->T; s0 += kFusedStepsPerLaunch) {
            g.step0 = s0;
            g.step1 = std::min(e->T, s0 + kFusedStepsPerLaunch);
            g.x0 = s0 == 0 ? e->bx0.d() : e->bfin.d();
            Scope sc(e, st, "dk_fused_filter<marginals>", e->profile != 0);
            if (e->Dp == 32) hipLaunchKernelGGL(dk_fused_filter<32>, dim3(1), dim3(256), FusedCfg<32>::LDS_BYTES, st, g);
            else if (e->Dp == 48) hipLaunchKernelGGL(dk_fused_filter<48>, dim3(1), dim3(256), FusedCfg<48>::LDS_BYTES, st, g);
            else hipLaunchKernelGGL(dk_fused_filter<64>, dim3(1), dim3(256), FusedCfg<64>::LDS_BYTES, st, g);
        }
        DCHK(hipStreamSynchronize(st));
        resolve(e);
        return TGP_OK;
    }
    const int Dp = e->Dp, Pq = e->Pq;
    const size_t DD = (size_t)Dp * Dp;
    (void)result8;
    DCHK(hipMemcpyAsync(e->bP.p, e->bx0.p, DD * 8, hipMemcpyDeviceToDevice, st));
    DCHK(hipMemcpyAsync(e->bm.p, e->bx0.d() + DD, (size_t)Dp * 8, hipMemcpyDeviceToDevice, st));
    Buf zero;
    DCHK(zero.ensure((size_t)e->p * 8));
    DCHK(hipMemsetAsync(zero.p, 0, (size_t)e->p * 8, st));
    for (int64_t step = 0; step < e->T; ++step) {
        const int64_t t = e->ordering == 0 ? step : e->T - 1 - step;
        const StepPtrs s = step_ptrs(e, t);
        auto emit = [&](const double* mx, const double* Px) {
            GemmArgs g;
            g.A = s.H; g.lda = Pq;
            g.B = Px; g.ldb = Dp;
            g.C = e->bV.d(); g.ldc = Pq;
            g.M = Pq; g.N = Dp; g.K = Dp;
            g.v.mode = 2;
            g.v.Mx = s.H; g.v.ld = Pq; g.v.n = Pq; g.v.K = Dp;
            g.v.x = mx; g.v.xs = 1;
            g.v.out = e->bV.d() + (size_t)Dp * Pq;
            g.v.y = zero.d(); g.v.mask = nullptr; g.v.hh = s.h; g.v.p = e->p;
            g.v.scal = e->bscal.d();
            launch_gemm(g, st);
            hipLaunchKernelGGL(dk_marg_diag, dim3((e->p + 15) / 16), dim3(256), 0, st, e->bV.d(), s.H, Pq, Dp, s.R,
                               e->bV.d() + (size_t)Dp * Pq, e->p, mean_out + t * e->p, var_out + t * e->p);
        };
        if (e->ordering == 0) {
            enqueue_predict(e, s, st, false);
            emit(e->bmp.d(), e->bPp.d());
        } else {
            emit(e->bm.d(), e->bP.d());
            enqueue_predict(e, s, st, false);
        }
        DCHK(hipMemcpyAsync(e->bP.p, e->bPp.p, DD * 8, hipMemcpyDeviceToDevice, st));
        DCHK(hipMemcpyAsync(e->bm.p, e->bmp.p, (size_t)Dp * 8, hipMemcpyDeviceToDevice, st));
        if ((step & 1023) == 1023) DCHK(hipStreamSynchronize(st));
    }
    DCHK(hipStreamSynchronize(st));
    zero.release();
    return TGP_OK;
}

// out[r] = add[r] + sign * sum_k M1[k ld1 + r] x1[k] + sum_k M2[k ld2 + r] x2[k] + sd(r) e[r],  sd = sqrt(var[r] + jit)
// (16 rows x 16 K-slices per workgroup, fixed summation order): the vector products of rand (lgssm.jl:81-91) and of the
// materialised posterior (g = mf - G mp, lgssm.jl:231-238).
struct Gemv2 {
    const double* M1 = nullptr; int64_t ld1 = 0; const double* x1 = nullptr; int K1 = 0; double sign = 1.0;
    const double* M2 = nullptr; int64_t ld2 = 0; const double* x2 = nullptr; int K2 = 0;
    const double* add = nullptr;
    const double* var = nullptr; const double* e = nullptr; double jit = 0.0; int nvar = 0;
    double* out = nullptr; int n = 0;        // rows computed (padded count: rows >= the matrices' true size read zeros)
    double* out2 = nullptr; int n2 = 0;      // unpadded copy
};
__global__ void __launch_bounds__(256) dk_gemv2(Gemv2 g) {
    __shared__ double part[16][17];
    const int row = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int r = blockIdx.x * 16 + row;
    double s = 0.0;
    if (r < g.n) {
        if (g.M1)
            for (int k = sl; k < g.K1; k += 16) s += g.sign * g.M1[(int64_t)k * g.ld1 + r] * g.x1[k];
        if (g.M2)
            for (int k = sl; k < g.K2; k += 16) s += g.M2[(int64_t)k * g.ld2 + r] * g.x2[k];
    }
    part[sl][row] = s;
    __syncthreads();
    if (sl == 0 && r < g.n) {
        double t = g.add ? g.add[r] : 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += part[q][row];
        if (g.var && r < g.nvar) t += sqrt(g.var[r] + g.jit) * g.e[r];
        g.out[r] = t;
        if (g.out2 && r < g.n2) g.out2[r] = t;
    }
}

__global__ void dk_identity(double* __restrict__ W, int n) {     // n x n identity, column-major, ld n
    const int64_t tot = (int64_t)n * n;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < tot; e += (int64_t)gridDim.x * blockDim.x) W[e] = (e / n == e % n) ? 1.0 : 0.0;
}

namespace {

constexpr int kBlk = 256;     // diagonal block of the blocked d x d factorisation = what dk_chol / dk_trsm handle in one launch

// emission marginals of the state (mx, Px): mean = H mx + h, var = diag(H Px H') + Rv  ->  mean_out / var_out (p values)
void enqueue_emit(Engine* e, const StepPtrs& s, const double* mx, const double* Px, const double* Rv, double* mean_out, double* var_out, hipStream_t st) {
    const int Dp = e->Dp, Pq = e->Pq;
    if (e->nnzH) {
        SpVec v;
        v.mode = 2; v.x = mx; v.out = e->bV.d() + (size_t)Dp * Pq;
        v.y = e->bzero.d(); v.mask = nullptr; v.hh = s.h; v.p = e->p; v.scal = e->bscal.d();
        hipLaunchKernelGGL(dk_spl<2>, dim3((Pq + 255) / 256, (Dp + 1) / 2 + 1), dim3(256), 0, st, static_cast<const int*>(e->bHcol.p), e->bHval.d(), e->nnzH,
                           Pq, Px, (int64_t)Dp, e->bV.d(), (int64_t)Pq, Dp, v);
    } else {
        GemmArgs g;
        g.A = s.H; g.lda = Pq;
        g.B = Px; g.ldb = Dp;
        g.C = e->bV.d(); g.ldc = Pq;
        g.M = Pq; g.N = Dp; g.K = Dp;
        g.v.mode = 2;
        g.v.Mx = s.H; g.v.ld = Pq; g.v.n = Pq; g.v.K = Dp;
        g.v.x = mx; g.v.xs = 1;
        g.v.out = e->bV.d() + (size_t)Dp * Pq;
        g.v.y = e->bzero.d(); g.v.mask = nullptr; g.v.hh = s.h; g.v.p = e->p;
        g.v.scal = e->bscal.d();
        launch_gemm(g, st);
    }
    hipLaunchKernelGGL(dk_marg_diag, dim3((e->p + 15) / 16), dim3(256), 0, st, e->bV.d(), s.H, Pq, Dp, Rv, e->bV.d() + (size_t)Dp * Pq, e->p, mean_out,
                       var_out);
}

// Blocked Cholesky of the Dp x Dp matrix Sm (column-major, ld Dp; overwritten by the trailing updates) + jitter I:
// L -> e->bLd (column-major, ld Dp), inverses of its 16 x 16 diagonal tiles -> e->bDinvd. Diagonal blocks of 256 by dk_chol,
// the panel below by dk_trsm (written in place: B = L_kk^-1 S_k,rest in row-major IS the column-major panel of L), the trailing
// update by dk_gemm.
void enqueue_chol_blocked(Engine* e, double* Sm, double jitter, hipStream_t st) {
    const int n = e->Dp;
    const int64_t ld = n;
    double* Lb = e->bLd.d();
    for (int k0 = 0; k0 < n; k0 += kBlk) {
        const int nb = std::min(kBlk, n - k0);
        const int* slots = static_cast<const int*>(nb == std::min(kBlk, n) ? e->bslots_blk.p : e->bslots_tail.p);
        hipLaunchKernelGGL(dk_chol, dim3(1), dim3(1024), 0, st, Sm + k0 + (int64_t)k0 * ld, ld, nb, jitter, slots, Lb + k0 + (int64_t)k0 * ld, ld,
                           e->bDinvd.d() + (size_t)(k0 / 16) * 256, e->bscal.d() + 4
#ifdef DK_TRACE
                           , (long long*)nullptr
#endif
        );
        const int rest = n - k0 - nb;
        if (rest <= 0) break;
        hipLaunchKernelGGL(dk_trsm, dim3(rest / 16), dim3(256), 0, st, Lb + k0 + (int64_t)k0 * ld, ld, e->bDinvd.d() + (size_t)(k0 / 16) * 256,
                           Sm + k0 + (int64_t)(k0 + nb) * ld, ld, nb, Lb + (k0 + nb) + (int64_t)k0 * ld, ld);
        GemmArgs g;     // S[rest, rest] -= L[rest, k] L[rest, k]'
        g.A = Lb + (k0 + nb) + (int64_t)k0 * ld; g.lda = ld;
        g.B = g.A; g.ldb = ld;
        g.C = Sm + (k0 + nb) + (int64_t)(k0 + nb) * ld; g.ldc = ld;
        g.E = g.C; g.lde = ld; g.sign = -1.0;
        g.M = g.N = rest; g.K = nb;
        launch_gemm(g, st);
    }
}

// X = L^-1 B with the blocked factor: B in the "V layout" (B[i][k] at Vb[i + k ldV], i < Dp, ncols columns; overwritten),
// X row-major (X[i][k] at Xm[i ldX + k]).
void enqueue_trsm_blocked(Engine* e, double* Vb, int64_t ldV, int ncols, double* Xm, int64_t ldX, hipStream_t st) {
    const int n = e->Dp;
    const int64_t ld = n;
    const double* Lb = e->bLd.d();
    for (int k0 = 0; k0 < n; k0 += kBlk) {
        const int nb = std::min(kBlk, n - k0);
        hipLaunchKernelGGL(dk_trsm, dim3(ncols / 16), dim3(256), 0, st, Lb + k0 + (int64_t)k0 * ld, ld, e->bDinvd.d() + (size_t)(k0 / 16) * 256, Vb + k0, ldV, nb,
                           Xm + (int64_t)k0 * ldX, ldX);
        const int rest = n - k0 - nb;
        if (rest <= 0) break;
        GemmArgs g;     // B[rest, :] -= L[rest, k] X[k, :]
        g.A = Lb + (k0 + nb) + (int64_t)k0 * ld; g.lda = ld;
        g.B = Xm + (int64_t)k0 * ldX; g.ldb = ldX;
        g.C = Vb + (k0 + nb); g.ldc = ldV;
        g.E = g.C; g.lde = ldV; g.sign = -1.0;
        g.M = rest; g.N = ncols; g.K = nb;
        launch_gemm(g, st);
    }
}

__global__ void dk_copy_with_column(const double* __restrict__ P, int Dp, const double* __restrict__ ms, const double* __restrict__ mp, double* __restrict__ W) {
    // W (Dp x (Dp + 16), ld Dp) = [P | ms - mp | 0 ...]
    const int64_t n = (int64_t)Dp * (Dp + 16);
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t col = e / Dp;
        const int i = (int)(e - col * Dp);
        W[e] = col < Dp ? P[e] : (col == Dp ? ms[i] - mp[i] : 0.0);
    }
}

// buffers and dk_chol slot tables of the blocked d x d factorisation (smoother, materialised posterior, rand)
int ensure_blocked_workspace(Engine* e, hipStream_t st) {
    const int Dp = e->Dp, Pq = e->Pq;
    const size_t DD = (size_t)Dp * Dp;
    const int64_t ldW = Dp + 16;
    DCHK(e->bLd.ensure(DD * 8));
    DCHK(e->bDinvd.ensure((size_t)(Dp / 16) * 256 * 8));
    for (Buf* b : {&e->bW0, &e->bW1, &e->bW2, &e->bW3}) DCHK(b->ensure((size_t)Dp * ldW * 8));
    DCHK(e->bzero.ensure((size_t)Pq * 8 + 64));
    DCHK(hipMemsetAsync(e->bzero.p, 0, (size_t)Pq * 8 + 64, st));
    DCHK(hipMemsetAsync(e->bLd.p, 0, DD * 8, st));
    DCHK(hipMemsetAsync(e->bscal.d() + 4, 0, 4 * sizeof(double), st));
    std::vector<int> tab, t2;
    chol_slot_table(std::min(kBlk, Dp) / 16, tab);
    DCHK(e->bslots_blk.ensure(tab.size() * sizeof(int)));
    DCHK(hipMemcpyAsync(e->bslots_blk.p, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice, st));
    const int tail = Dp > kBlk ? Dp % kBlk : 0;
    if (tail) {
        chol_slot_table(tail / 16, t2);
        DCHK(e->bslots_tail.ensure(t2.size() * sizeof(int)));
        DCHK(hipMemcpyAsync(e->bslots_tail.p, t2.data(), t2.size() * sizeof(int), hipMemcpyHostToDevice, st));
    }
    DCHK(hipStreamSynchronize(st));     // the tables are host vectors on this frame
    return TGP_OK;
}
int blocked_chol_status(Engine* e, const char* what) {
    double flag[4] = {0, 0, 0, 0};
    DCHK(hipMemcpy(flag, e->bscal.d() + 4, 4 * sizeof(double), hipMemcpyDeviceToHost));
    if (flag[3] != 0.0) return e->fail(TGP_ENOTPD, what);
    return TGP_OK;
}

// One backward step: (bm, bP) = smoothed state of step t  ->  smoothed state of step t - 1, given the filtering state
// (mf, Pf) of step t - 1 and step t's transition.
int smoother_move(Engine* e, int64_t t, const double* Pf, const double* mf, hipStream_t st, bool prof) {
    const int Dp = e->Dp;
    const size_t DD = (size_t)Dp * Dp;
    const int64_t ldW = Dp + 16;
    const StepPtrs s = step_ptrs(e, t);
    // predict from the filtering state of step t-1 with step t's transition: T1 = A Pf, Pp = T1 A' + Q, mp = A mf + a
    DCHK(hipMemcpyAsync(e->bW3.p, e->bP.p, DD * 8, hipMemcpyDeviceToDevice, st));            // keep Ps
    DCHK(hipMemcpyAsync(e->bW2.p, e->bm.p, (size_t)Dp * 8, hipMemcpyDeviceToDevice, st));    // keep ms (first Dp doubles of W2)
    DCHK(hipMemcpyAsync(e->bP.p, Pf, DD * 8, hipMemcpyDeviceToDevice, st));
    DCHK(hipMemcpyAsync(e->bm.p, mf, (size_t)Dp * 8, hipMemcpyDeviceToDevice, st));
    enqueue_predict(e, s, st, prof);                   // bT1 = A Pf, bPp, bmp
    hipLaunchKernelGGL(dk_copy_with_column, dim3(512), dim3(256), 0, st, e->bW3.d(), Dp, e->bW2.d(), e->bmp.d(), e->bW0.d());   // W0 = [Ps | ms - mp | 0]
    {
        Scope sc(e, st, "smoother: blocked chol(Pp)", prof);
        enqueue_chol_blocked(e, e->bPp.d(), 1e-10, st);    // Lc Lc' = Pp + 1e-10 I (lgssm.jl:235)
    }
    {
        Scope sc(e, st, "smoother: 3 blocked trsm", prof);
        enqueue_trsm_blocked(e, e->bT1.d(), Dp, Dp, e->bW1.d(), ldW, st);            // Z  = Lc^-1 (A Pf)            -> W1 (row-major)
        enqueue_trsm_blocked(e, e->bW0.d(), Dp, Dp + 16, e->bW2.d(), ldW, st);       // Y  = Lc^-1 [Ps | ms - mp]    -> W2; u = column Dp
        enqueue_trsm_blocked(e, e->bW2.d(), ldW, Dp, e->bW3.d(), ldW, st);           // W  = Lc^-1 Y'                -> W3
    }
    Scope sc(e, st, "smoother: 2 dk_gemm (Pf + Z'(W - I)Z)", prof);
    {   // C' = Z' - Z' W  (= -((W - I) Z)')  -> bT1 (column-major, ld Dp)
        GemmArgs g;
        g.A = e->bW1.d(); g.lda = ldW;
        g.B = e->bW3.d(); g.ldb = ldW;
        g.C = e->bT1.d(); g.ldc = Dp;
        g.E = e->bW1.d(); g.lde = ldW; g.sign = -1.0;
        g.M = g.N = g.K = Dp;
        launch_gemm(g, st);
    }
    {   // Ps <- Pf - Z' (-(W - I) Z) = G Ps G' + L ; rider: ms <- mf + Z' u
        GemmArgs g;
        g.A = e->bW1.d(); g.lda = ldW;
        g.B = e->bT1.d(); g.ldb = Dp;
        g.C = e->bP.d(); g.ldc = Dp;
        g.E = Pf; g.lde = Dp; g.sign = -1.0;
        g.M = g.N = g.K = Dp;
        g.v.mode = 1;
        g.v.Mx = e->bW1.d(); g.v.ld = ldW; g.v.n = Dp; g.v.K = Dp;
        g.v.x = e->bW2.d() + Dp; g.v.xs = ldW; g.v.add = mf; g.v.out = e->bm.d();
        launch_gemm(g, st);
    }
    return TGP_OK;
}

}  // namespace

// marginals(replace_observation_noise_cov(posterior(model, y), Rnew)) for the dense path: forward filter keeping the filtering
// states, then per step the reference's invert_dynamics + Reverse step_marginals (lgssm.jl:111-115, 215-238) with
//   Lc Lc' = Pp + 1e-10 I,  Z = Lc^-1 (A Pf),  G = Z' Lc^-1,  L = Pf - Z'Z,
//   x <- G x + g = mf + Z' Lc^-1 (ms - mp),   P <- G Ps G' + L = Pf + Z' (W - I) Z  with  W = Lc^-1 Ps Lc^-T
// so that only FORWARD substitutions with the blocked factor and MFMA GEMMs are needed.
int posterior_marginals(Engine* e, const double* y, const uint8_t* mask, const double* Rnew, int64_t sRn, double* mean_out, double* var_out,
                        double* result8, hipStream_t st) {
    if (!e->have_model) return e->fail(TGP_EINVAL, "no model");
    if (e->ordering != 0) return e->fail(TGP_EUNSUPPORTED, "dense path: posterior of a Reverse-ordered model is not implemented");
    DCHK(hipSetDevice(e->device));
    // The persistent backward pass is a modified Bryson-Frazier recursion: it has no counterpart of the reference's 1e-10 jitter on the
    // predicted covariance in invert_dynamics (lgssm.jl:235) and forms variances by a difference, so it agrees with the reference's RTS
    // chain to ~1e-6 relative only. It is therefore opt-in (TGP_OPT_DENSE_FUSED = 2); the default is the jitter-faithful chain below.
    if (fused(e) && e->fused_opt >= 2 && e->segment_opt == 0) {
        const int rcq = fused_posterior_marginals(e, y, mask, Rnew, sRn ? 1 : 0, mean_out, var_out, result8, st);
        if (rcq != TGP_EUNSUPPORTED) return rcq;
    }
    const int Dp = e->Dp, Pq = e->Pq;
    const size_t DD = (size_t)Dp * Dp;
    const int64_t ldW = Dp + 16;
    const int64_t T = e->T;
    // Storage of the filtering states the backward pass reads. All T of them when they fit in a quarter of the free memory;
    // otherwise segments of S ~ sqrt(T) steps: pass 1 keeps the state at every segment boundary, the backward pass re-filters
    // one segment at a time from its boundary (bit-identical: the kernels reduce in a fixed order), 2 filters + 1 smoother.
    size_t free_b = 0, total_b = 0;
    DCHK(hipMemGetInfo(&free_b, &total_b));
    const double avail = (double)free_b + (double)e->bPstore.cap + (double)e->bmstore.cap + (double)e->bPbound.cap + (double)e->bmbound.cap;
    const double per_step = (double)(DD + Dp) * 8.0;
    int64_t S = T;
    if (e->segment_opt > 0) S = std::min<int64_t>(T, e->segment_opt);
    else if ((double)T * per_step > 0.25 * avail) S = std::max<int64_t>(16, (int64_t)std::ceil(std::sqrt((double)T)));
    const int64_t nseg = (T + S - 1) / S;
    if ((double)(S + nseg) * per_step > 0.8 * avail)
        return e->fail(TGP_EUNSUPPORTED, "dense path: the smoother's stored filtering covariances (2 sqrt(T) d^2 doubles) do not fit in this GPU's free memory");
    DCHK(e->bPstore.ensure((size_t)S * DD * 8));
    DCHK(e->bmstore.ensure((size_t)S * Dp * 8));
    DCHK(e->bPbound.ensure((size_t)nseg * DD * 8));
    DCHK(e->bmbound.ensure((size_t)nseg * Dp * 8));
    {
        const int rcw = ensure_blocked_workspace(e, st);
        if (rcw != TGP_OK) return rcw;
    }
    double* scratch8 = e->bzero.d() + Pq;       // lml / flags of the re-filtered segments (already counted in pass 1)
    // filter steps [t0, t1) from the state in (bm, bP); keep every state of the segment when `keep`
    auto filter_range = [&](int64_t t0, int64_t t1, bool keep, double* res) -> int {
        for (int64_t t = t0; t < t1; ++t) {
            const StepPtrs s = step_ptrs(e, t);
            const bool prof = e->profile && res == result8 && (t % 16 == 8 || T < 64);
            enqueue_predict(e, s, st, prof);
            enqueue_update(e, s, t, y, mask, keep ? e->bmstore.d() - t0 * Dp : nullptr, keep ? e->bPstore.d() - t0 * (int64_t)DD : nullptr, res, st, prof,
                           Dp);
            if ((t & 1023) == 1023) {
                DCHK(hipStreamSynchronize(st));
                resolve(e);
            }
        }
        return TGP_OK;
    };
    // ---- pass 1: filter (lml into result8), keeping the state that enters every segment; the last segment is kept in full
    DCHK(hipMemcpyAsync(e->bP.p, e->bx0.p, DD * 8, hipMemcpyDeviceToDevice, st));
    DCHK(hipMemcpyAsync(e->bm.p, e->bx0.d() + DD, (size_t)Dp * 8, hipMemcpyDeviceToDevice, st));
    for (int64_t seg = 0; seg < nseg; ++seg) {
        DCHK(hipMemcpyAsync(e->bPbound.d() + seg * DD, e->bP.p, DD * 8, hipMemcpyDeviceToDevice, st));
        DCHK(hipMemcpyAsync(e->bmbound.d() + seg * Dp, e->bm.p, (size_t)Dp * 8, hipMemcpyDeviceToDevice, st));
        const int rcf = filter_range(seg * S, std::min(T, (seg + 1) * S), seg == nseg - 1, result8);
        if (rcf != TGP_OK) return rcf;
    }
    // ---- backward: (bm, bP) hold the smoothed state of step t
    for (int64_t seg = nseg - 1; seg >= 0; --seg) {
        const int64_t t0 = seg * S, t1 = std::min(T, (seg + 1) * S);
        if (seg != nseg - 1) {
            // re-filter this segment from its boundary state; the smoothed state of step t1 - 1 waits in (W3, W2)
            DCHK(hipMemcpyAsync(e->bW3.p, e->bP.p, DD * 8, hipMemcpyDeviceToDevice, st));
            DCHK(hipMemcpyAsync(e->bW2.p, e->bm.p, (size_t)Dp * 8, hipMemcpyDeviceToDevice, st));
            DCHK(hipMemcpyAsync(e->bP.p, e->bPbound.d() + seg * DD, DD * 8, hipMemcpyDeviceToDevice, st));
            DCHK(hipMemcpyAsync(e->bm.p, e->bmbound.d() + seg * Dp, (size_t)Dp * 8, hipMemcpyDeviceToDevice, st));
            const int rcf = filter_range(t0, t1, true, scratch8);
            if (rcf != TGP_OK) return rcf;
            DCHK(hipMemcpyAsync(e->bP.p, e->bW3.p, DD * 8, hipMemcpyDeviceToDevice, st));
            DCHK(hipMemcpyAsync(e->bm.p, e->bW2.p, (size_t)Dp * 8, hipMemcpyDeviceToDevice, st));
        }
        for (int64_t t = t1 - 1; t >= t0; --t) {
            const bool prof = e->profile && (t % 16 == 8 || T < 64);
            {
                const StepPtrs se = step_ptrs(e, t);
                Scope sc(e, st, "smoother: emission marginals", prof);
                enqueue_emit(e, se, e->bm.d(), e->bP.d(), Rnew + (sRn ? t * e->p : 0), mean_out + t * e->p, var_out + t * e->p, st);
            }
            if (t == 0) break;
            // filtering state of step t - 1: inside this segment's store, or the state that entered the segment
            const double* Pf = t > t0 ? e->bPstore.d() + (size_t)(t - 1 - t0) * DD : e->bPbound.d() + (size_t)seg * DD;
            const double* mf = t > t0 ? e->bmstore.d() + (size_t)(t - 1 - t0) * Dp : e->bmbound.d() + (size_t)seg * Dp;
            const int rcm = smoother_move(e, t, Pf, mf, st, prof);
            if (rcm != TGP_OK) return rcm;
            if ((t & 255) == 255) {
                DCHK(hipStreamSynchronize(st));
                resolve(e);
            }
        }
    }
    int rc = TGP_OK;
    if (hipStreamSynchronize(st) != hipSuccess) rc = e->fail(TGP_EHIP, "dense smoother: stream error");
    resolve(e);
    if (rc == TGP_OK) rc = blocked_chol_status(e, "dense smoother: predicted covariance not positive definite (lgssm.jl:235)");
    return rc;
}

// posterior(prior, y) evaluated (lgssm.jl:193-238, Forward priors): per step the time-reversed transition
//   G = Pf A' (Pp + 1e-10 I)^-1 = Z' Lc^-1,  g = mf - G mp,  L = Pf - Z'Z,   Z = Lc^-1 (A Pf),  Lc Lc' = Pp + 1e-10 I
// (column-major d x d blocks / d vectors per step) and the final filtering state. Lc^-1 is formed explicitly (one blocked forward
// substitution against the identity), so that G is an MFMA GEMM instead of a backward substitution.
int posterior(Engine* e, const double* y, const uint8_t* mask, double* G_out, double* g_out, double* L_out, double* xfm_host, double* xfP_host,
              double* result8, hipStream_t st) {
    if (!e->have_model) return e->fail(TGP_EINVAL, "no model");
    if (e->ordering != 0) return e->fail(TGP_EUNSUPPORTED, "dense path: posterior of a Reverse-ordered model is not implemented");
    DCHK(hipSetDevice(e->device));
    const int Dp = e->Dp, d = e->d;
    const size_t DD = (size_t)Dp * Dp;
    const int64_t ldW = Dp + 16;
    {
        const int rcw = ensure_blocked_workspace(e, st);
        if (rcw != TGP_OK) return rcw;
    }
    DCHK(hipMemcpyAsync(e->bP.p, e->bx0.p, DD * 8, hipMemcpyDeviceToDevice, st));
    DCHK(hipMemcpyAsync(e->bm.p, e->bx0.d() + DD, (size_t)Dp * 8, hipMemcpyDeviceToDevice, st));
    for (int64_t t = 0; t < e->T; ++t) {
        const StepPtrs s = step_ptrs(e, t);
        const bool prof = e->profile && (t % 16 == 8 || e->T < 64);
        enqueue_predict(e, s, st, prof);                                                             // bT1 = A Pf, bPp, bmp
        if (G_out) {
            Scope sc(e, st, "posterior: G, g, L of the step", prof);
            DCHK(hipMemcpyAsync(e->bW0.p, e->bPp.p, DD * 8, hipMemcpyDeviceToDevice, st));
            enqueue_chol_blocked(e, e->bW0.d(), 1e-10, st);                                          // Lc (lgssm.jl:235)
            enqueue_trsm_blocked(e, e->bT1.d(), Dp, Dp, e->bW1.d(), ldW, st);                        // Z -> W1 (row-major)
            hipLaunchKernelGGL(dk_identity, dim3(512), dim3(256), 0, st, e->bW2.d(), Dp);
            enqueue_trsm_blocked(e, e->bW2.d(), Dp, Dp, e->bW3.d(), ldW, st);                        // Lc^-1 -> W3 (row-major)
            {   // G = Z' Lc^-1 -> bT1 (padded, column-major) and G_out[t]
                GemmArgs g;
                g.A = e->bW1.d(); g.lda = ldW;
                g.B = e->bW3.d(); g.ldb = ldW;
                g.C = e->bT1.d(); g.ldc = Dp;
                g.M = g.N = g.K = Dp;
                g.C2 = G_out + t * (int64_t)d * d; g.ldc2 = d; g.M2 = g.N2 = d;
                launch_gemm(g, st);
            }
            {   // L = Pf - Z'Z -> L_out[t]
                GemmArgs g;
                g.A = e->bW1.d(); g.lda = ldW;
                g.B = e->bW1.d(); g.ldb = ldW;
                g.C = e->bW0.d(); g.ldc = Dp;
                g.E = e->bP.d(); g.lde = Dp; g.sign = -1.0;
                g.M = g.N = g.K = Dp;
                g.C2 = L_out + t * (int64_t)d * d; g.ldc2 = d; g.M2 = g.N2 = d;
                launch_gemm(g, st);
            }
            Gemv2 v;     // g = mf - G mp
            v.M1 = e->bT1.d(); v.ld1 = Dp; v.x1 = e->bmp.d(); v.K1 = Dp; v.sign = -1.0;
            v.add = e->bm.d();
            v.out = e->bW2.d(); v.n = Dp;
            v.out2 = g_out + t * d; v.n2 = d;
            hipLaunchKernelGGL(dk_gemv2, dim3(Dp / 16), dim3(256), 0, st, v);
        }
        enqueue_update(e, s, t, y, mask, nullptr, nullptr, result8, st, prof);
        if ((t & 255) == 255) {
            DCHK(hipStreamSynchronize(st));
            resolve(e);
        }
    }
    DCHK(hipStreamSynchronize(st));
    resolve(e);
    if (xfm_host && xfP_host) {
        std::vector<double> hP(DD), hm(Dp);
        DCHK(hipMemcpy(hP.data(), e->bP.p, DD * 8, hipMemcpyDeviceToHost));
        DCHK(hipMemcpy(hm.data(), e->bm.p, (size_t)Dp * 8, hipMemcpyDeviceToHost));
        for (int j = 0; j < d; ++j) {
            xfm_host[j] = hm[j];
            for (int i = 0; i < d; ++i) xfP_host[i + (size_t)j * d] = hP[i + (size_t)j * Dp];
        }
    }
    return G_out ? blocked_chol_status(e, "dense posterior: predicted covariance not positive definite (lgssm.jl:235)") : TGP_OK;
}

// rand(rng, model) with the randomness supplied (lgssm.jl:65-91): x <- A x + a + chol(Q + 1e-9 I).L eps_t (lgc.jl:84-87),
// y = H x + h + sqrt(R (+ 1e-9 for SmallOutputLGC)) .* eps_e (lgc.jl:84-87 with diagonal R / :241-243). x0 (drawn on the host,
// gaussian.jl:35-43) comes in as a d-vector. One factorisation of Q when it is shared, one per step otherwise.
int rand(Engine* e, const double* x0_host, const double* eps_t, const double* eps_e, int small_out, double* y_out, hipStream_t st) {
    if (!e->have_model) return e->fail(TGP_EINVAL, "no model");
    DCHK(hipSetDevice(e->device));
    const int Dp = e->Dp, Pq = e->Pq, d = e->d, p = e->p;
    const size_t DD = (size_t)Dp * Dp;
    {
        const int rcw = ensure_blocked_workspace(e, st);
        if (rcw != TGP_OK) return rcw;
    }
    std::vector<double> xh(Dp, 0.0);
    for (int i = 0; i < d; ++i) xh[i] = x0_host[i];
    DCHK(hipMemcpyAsync(e->bm.p, xh.data(), (size_t)Dp * 8, hipMemcpyHostToDevice, st));
    DCHK(hipStreamSynchronize(st));
    auto factor_Q = [&](const double* Qt) -> int {
        DCHK(hipMemcpyAsync(e->bW0.p, Qt, DD * 8, hipMemcpyDeviceToDevice, st));
        enqueue_chol_blocked(e, e->bW0.d(), 1e-9, st);
        return TGP_OK;
    };
    if (e->sQ == 0) {
        const int rcq = factor_Q(e->bQ.d());
        if (rcq != TGP_OK) return rcq;
    }
    if (fused(e) && e->sA == 0 && e->sa == 0 && e->sQ == 0) {      // mid-sized state, shared transition: one persistent kernel
        FusedRandArgs g;
        g.T = e->T; g.d = d; g.p = p; g.Pq = Pq; g.ordering = e->ordering; g.small_out = small_out;
        g.A = e->bA.d(); g.Lq = e->bLd.d(); g.a = e->ba.d(); g.H = e->bH.d(); g.h = e->bh.d(); g.R = e->bR.d();
        g.sH = e->sH; g.sh = e->sh; g.sR = e->sR;
        g.x0 = e->bm.d(); g.eps_t = eps_t; g.eps_e = eps_e; g.y_out = y_out;
        {
            Scope sc(e, st, "dk_fused_rand", e->profile != 0);
            if (Dp == 32) hipLaunchKernelGGL(dk_fused_rand<32>, dim3(1), dim3(256), FusedRandCfg<32>::LDS_BYTES, st, g);
            else if (Dp == 48) hipLaunchKernelGGL(dk_fused_rand<48>, dim3(1), dim3(256), FusedRandCfg<48>::LDS_BYTES, st, g);
            else hipLaunchKernelGGL(dk_fused_rand<64>, dim3(1), dim3(256), FusedRandCfg<64>::LDS_BYTES, st, g);
        }
        DCHK(hipStreamSynchronize(st));
        resolve(e);
        return blocked_chol_status(e, "dense rand: Q + 1e-9 I is not positive definite (lgc.jl:86)");
    }
    for (int64_t step = 0; step < e->T; ++step) {
        const int64_t t = e->ordering == 0 ? step : e->T - 1 - step;
        const StepPtrs s = step_ptrs(e, t);
        auto emit = [&](const double* x) {
            Gemv2 v;
            v.M1 = s.H; v.ld1 = Pq; v.x1 = x; v.K1 = Dp;
            v.add = s.h;
            v.var = s.R; v.e = eps_e + t * p; v.jit = small_out ? 1e-9 : 0.0; v.nvar = p;
            v.out = e->bV.d(); v.n = p;
            v.out2 = y_out + t * p; v.n2 = p;
            hipLaunchKernelGGL(dk_gemv2, dim3((p + 15) / 16), dim3(256), 0, st, v);
        };
        auto move = [&]() -> int {
            if (e->sQ != 0) {
                const int rcq = factor_Q(s.Q);
                if (rcq != TGP_OK) return rcq;
            }
            Gemv2 v;
            v.M1 = s.A; v.ld1 = Dp; v.x1 = e->bm.d(); v.K1 = Dp;
            v.M2 = e->bLd.d(); v.ld2 = Dp; v.x2 = eps_t + t * d; v.K2 = d;
            v.add = s.a;
            v.out = e->bmp.d(); v.n = Dp;
            hipLaunchKernelGGL(dk_gemv2, dim3(Dp / 16), dim3(256), 0, st, v);
            std::swap(e->bm.p, e->bmp.p);
            std::swap(e->bm.cap, e->bmp.cap);
            return TGP_OK;
        };
        if (e->ordering == 0) {
            const int rcm = move();
            if (rcm != TGP_OK) return rcm;
            emit(e->bm.d());
        } else {
            emit(e->bm.d());
            const int rcm = move();
            if (rcm != TGP_OK) return rcm;
        }
        if ((step & 1023) == 1023) DCHK(hipStreamSynchronize(st));
    }
    DCHK(hipStreamSynchronize(st));
    return blocked_chol_status(e, "dense rand: Q + 1e-9 I is not positive definite (lgc.jl:86)");
}

}  // namespace tgp_dense
