// Streaming logpdf kernel of the stationary-gain engine (round 6) -- see tgp_lml.hpp.  gfx950 only (wave64, DPP scans, 160 KB of LDS per CU,
// uniform coefficients through the kernel-argument segment).
#include "tgp_lml.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace tgp_lml {

namespace {

typedef double v2d __attribute__((ext_vector_type(2)));

// ---- kernel arguments: every coefficient is wave-uniform and reaches the lanes through scalar loads ---------------------------------
template <int D, int N>
struct LArgs {
    double fd[D], fo[D], fb[D], fc[D], fw[D];      // z' = fd z + fo z_partner + fb y + fc (fc = fa - fb hh),  r = (y - hh) - fw . z
    double hh;
    double lvr[6][D], lvi[6][D];                   // M^(N 2^k), k < 6: the block form (re, signed im)
    double wj[N][D];                               // fw' M^j (a row): step j of a lane sees the lane's start state through it
    double WN[D][D];                               // sum_{j < N} w_j' w_j: what a lane's start state adds to its sum of squares
    double Wt[D][D];                               // sum_t w_t' w_t over a run's first tile
    long long T, nhs, G, W, Chi, Clo;      // G tiles behind the head, W of them whole; a workgroup's waves 0-3 run Chi tiles each, its waves 4-7 Clo
    const double* y;
    double* part;
    double* head_in;
    long long* flags;
    long long seq;
    int nwg;
    int dbg;            // development: 1 = wave 0 of every workgroup stamps its phases (100 MHz) into part[kStampOff + 8 wg ...]
};

template <int D>
__device__ __forceinline__ constexpr int partner(int i) {
    return ((i ^ 1) < D) ? (i ^ 1) : i;
}

__device__ __forceinline__ void lds_sync() {      // one wave talking to itself through LDS (DS operations of a wave execute in order)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ double readlane_d(double x, int l) {      // l wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), l), hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
    return __hiloint2double(hi, lo);
}
// DPP moves of a double (two v_mov_dpp; lanes without a source and rows outside the mask read zero).  gfx9 controls: row_shr:n 0x110 + n,
// wave_shr:1 0x138, row_bcast:15 0x142, row_bcast:31 0x143
template <int CTRL, int ROWMASK = 0xF>
__device__ __forceinline__ double dpp_mov(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROWMASK, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROWMASK, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double x) {      // the sum over the wave, in every lane's SGPR copy (lane 63 holds it)
    x += dpp_mov<0x111>(x);
    x += dpp_mov<0x112>(x);
    x += dpp_mov<0x114>(x);
    x += dpp_mov<0x118>(x);
    x += dpp_mov<0x142, 0xA>(x);
    x += dpp_mov<0x143, 0xC>(x);
    return readlane_d(x, 63);
}

// the 16-byte slot of piece j of lane L in a wave's LDS slice: PPL pieces per lane, XOR-swizzled so that the sixteen lanes a ds_read_b128
// serves in one LDS cycle ({0-3, 12-15, 20-27}, ...) fall on the sixteen slots of a 256-byte bank row, and the eight consecutive lanes of a
// ds_write_b128 group (consecutive pieces of ONE lane's data) on eight distinct slots of their 128-byte window
template <int PPL>
__device__ __forceinline__ int slot_of(int L, int j) {
    static_assert(PPL == 8 || PPL == 16, "sixteen or thirty-two steps per lane");
    return L * PPL + (j ^ ((L / (16 / PPL)) & (PPL - 1)));
}

template <int D, int N, bool ALIGNED>
__global__ __launch_bounds__(kNW * 64, (N == 16 ? 4 : 2)) void k_lml_stream(const LArgs<D, N> by_value) {
    // (read where they lie, in the kernel-argument segment: tgp_modal.hip k_steady_one on why)
    (void)by_value;
    // (the pointer keeps its address space -- constant, 4 -- through the optimisation barrier in the tile loop: scalar loads, not flat ones)
    typedef const __attribute__((address_space(4))) LArgs<D, N>* KaPtr;
    KaPtr kap = (KaPtr)__builtin_amdgcn_kernarg_segment_ptr();
#define ka (*kap)
    {
        // Touch every 64-byte line of the argument segment in ONE batch of scalar loads: the segment lives in host memory, a cold line is a PCIe round
        // trip (~2 us), and the optimisation barriers below make the waves read the coefficients phase by phase -- five or six cold batches in a
        // row were ~8 us of every wave's life, the whole launch at small T (k_steady_one reads its arguments in one batch by construction)
        typedef const __attribute__((address_space(4))) unsigned* WordPtr;
        WordPtr w = (WordPtr)kap;
        unsigned warm = 0u;
#pragma unroll
        for (unsigned o = 0; o < sizeof(LArgs<D, N>); o += 64) warm ^= w[o / 4];
        asm volatile("" ::"s"(warm));
    }
    constexpr int PPL = N / 2, TILE = 64 * N;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    v2d* sY = reinterpret_cast<v2d*>(lds_raw) + (size_t)wave * (64 * PPL);
    double* sPost = reinterpret_cast<double*>(lds_raw + (size_t)kNW * 64 * PPL * 16);      // [kNW][2 + 2 D]: Q, active, V, E per wave
    constexpr int PW = 2 + 2 * D;

    // The run's tiles.  Waves w and w + 4 of a workgroup share a SIMD: the first four run Chi tiles each, the last four Clo (Chi - Clo <= 1), so that
    // every SIMD of the chip holds Chi + Clo tiles -- with equal runs of three at T = 1e7 the launch was 204 workgroups on 256 CUs, six tiles on the busy
    // SIMDs and none on a fifth of them; as (3, 2) it is 245 workgroups and five tiles everywhere.  (Dealt out without regard to the SIMDs, runs of
    // two and three had the three-tile runs end 7 us behind the others.)
    const long long g0 = (long long)blockIdx.x * 4 * (ka.Chi + ka.Clo) + (wave < 4 ? wave * ka.Chi : 4 * ka.Chi + (wave - 4) * ka.Clo);
    const long long g1_own = g0 + (wave < 4 ? ka.Chi : ka.Clo);
    // (a run holds at least one WHOLE tile -- what its first-tile closure sums over -- unless the series is shorter than a tile: one run, one partial tile)
    const bool active = ka.W > 0 ? (g0 < ka.W && g1_own > g0) : (blockIdx.x == 0 && wave == 0);
    const bool stamp = ka.dbg && wave == 0 && lane == 0 && blockIdx.x < 512;
    if (stamp) ka.part[kStampOff + 8 * blockIdx.x + 0] = (double)wall_clock64();
    // the head's observations to the host, first thing (its forward recursion runs there, beside the kernel)
    // The head's observations to the host (its forward recursion runs there, beside the kernel): write-through stores now, the flag at the END of this
    // wave's run -- the stores' acknowledgement from host memory takes ~7 us (measured: the workgroup's barrier stood that long behind this wave when
    // it waited for it up here); a head wave without a run (a short series) raises no flag at all: the kernel's end delivers the data
    const bool head_wave = blockIdx.x == 0 && wave == kNW - 1;
    bool head_flag_due = false;
    if (head_wave) {
        for (int t = lane; t < (int)ka.nhs; t += 64) {
            const double v = ka.y[t];
            double* dst = ka.head_in + t;
            asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(dst), "v"(v) : "memory");
        }
        head_flag_due = true;
    }
    auto raise_head_flag = [&]() {      // (every vector-memory operation of the wave has returned when this is called)
        if (lane == 0) {
            const long long fv = 2 * ka.seq;
            long long* fp = ka.flags;
            asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(fp), "v"(fv) : "memory");
        }
        head_flag_due = false;
    };
    double Q = 0.0, V[D], E[D];
#pragma unroll
    for (int i = 0; i < D; ++i) V[i] = E[i] = 0.0;
    if (active) {
        // (only the series' last tile can be a partial one: it belongs to the run that holds the last whole tile)
        const long long g1 = g1_own >= ka.W ? ka.G : g1_own;
        const long long t_lo = ka.nhs + g0 * TILE;
        long long t_hi = ka.nhs + g1 * TILE;
        t_hi = t_hi < ka.T ? t_hi : ka.T;
        const double* __restrict__ y = ka.y;
        // M^(N e), e the bits of `ex` (block form), from the levels M^(N 2^k)
        auto lane_power = [&](int ex, double (&pr)[D], double (&pi)[D]) {
#pragma unroll
            for (int i = 0; i < D; ++i) {
                double xr = 1.0, xi = 0.0;
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const double lr = ka.lvr[k][i], li = ka.lvi[k][i];
                    const bool bit = ((ex >> k) & 1) != 0;
                    const double nr = fma(xr, lr, -(xi * li)), ni = fma(xr, li, xi * lr);
                    xr = bit ? nr : xr;
                    xi = bit ? ni : xi;
                }
                pr[i] = xr;
                pi[i] = xi;
            }
        };
        double mpr[D], mpi[D];      // M^(N (p + 1)), p the lane's place in its row of sixteen: what carries a row's entering state to the lane
        lane_power((lane & 15) + 1, mpr, mpi);

        v2d stage[PPL];
        // piece q = 64 k + lane of a WHOLE tile: the steps tile_t0 + 2 q, + 1 (every load instruction 1 KB of consecutive bytes)
        auto issue_loads = [&](long long tile_t0) {
            if (ALIGNED) {
                const v2d* __restrict__ src = reinterpret_cast<const v2d*>(y + tile_t0);
#pragma unroll
                for (int k = 0; k < PPL; ++k) stage[k] = __builtin_nontemporal_load(src + k * 64 + lane);
            } else {      // a pointer off the 16-byte boundary
#pragma unroll
                for (int k = 0; k < PPL; ++k) {
                    const double* __restrict__ src = y + tile_t0 + 2 * (k * 64 + lane);
                    v2d w;
                    w.x = src[0];
                    w.y = src[1];
                    stage[k] = w;
                }
            }
        };
        // the series' last tile when it is a partial one: straight into the slice, steps beyond the end read as zero (one wave of the launch, once)
        auto load_tail = [&](long long tile_t0) {
#pragma unroll 1
            for (int k = 0; k < PPL; ++k) {
                const int q = k * 64 + lane;
                const long long t = tile_t0 + 2 * q;
                v2d w;
                w.x = t < t_hi ? y[t] : 0.0;
                w.y = t + 1 < t_hi ? y[t + 1] : 0.0;
                sY[slot_of<PPL>(q / PPL, q % PPL)] = w;
            }
        };
        auto stage_to_lds = [&]() {
#pragma unroll
            for (int k = 0; k < PPL; ++k) {
                const int q = k * 64 + lane;
                sY[slot_of<PPL>(q / PPL, q % PPL)] = stage[k];
            }
        };

        double zin[D], acc = 0.0;
#pragma unroll
        for (int i = 0; i < D; ++i) zin[i] = 0.0;
        bool staged = t_lo + TILE <= t_hi;      // the tile about to be worked on travels through `stage` (a whole tile) -- wave-uniform
        if (staged) issue_loads(t_lo);
        bool first_tile = true;
        for (long long tile_t0 = t_lo; tile_t0 < t_hi; tile_t0 += TILE) {
            // (the coefficients are re-read through the scalar cache every tile: hoisted out of the loop they would not fit the SGPRs)
            asm volatile("" : "+s"(kap));
            if (staged) stage_to_lds();      // (waits for the loads)
            else load_tail(tile_t0);
            lds_sync();
            if (stamp && first_tile) ka.part[kStampOff + 8 * blockIdx.x + 1] = (double)wall_clock64();
            // the two waves of a SIMD take turns at the higher priority, tile by tile (tgp_post.hip: the arbiter prefers the older wave)
            if ((((int)((tile_t0 - t_lo) / TILE)) ^ (wave >> 2)) & 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
            staged = tile_t0 + 2 * TILE <= t_hi;
            if (staged) issue_loads(tile_t0 + TILE);      // the next tile's observations travel while this one is in work
            // ---- ONE sweep from a zero start (a whole tile): the innovations r0 of the lane's steps are never kept -- what the lane's true
            // start state st makes of them is a quadratic form,  sum_j (r0_j - w_j . st)^2 = q - 2 st . v + st' WN st  with  q = sum r0^2,
            // v = sum_j w_j r0_j (w_j = fw' M^j from the argument table, WN = sum_j w_j' w_j data-free) -- 17 instructions per step at d = 3 where the
            // two-sweep form below spends 23.  A tile with steps beyond the series' end (one per launch at most) takes the two sweeps.
            constexpr int CH = N == 16 ? 4 : 8;
            const long long nv = t_hi - tile_t0;      // valid steps of the tile (wave-uniform)
            const bool full = nv >= TILE;
            const long long left_steps = nv - (long long)lane * N;
            const int nvalid = left_steps >= N ? N : (left_steps > 0 ? (int)left_steps : 0);
            double z[D], q = 0.0, v[D];
#pragma unroll
            for (int i = 0; i < D; ++i) z[i] = v[i] = 0.0;
#pragma unroll 1
            for (int jc = 0; jc < N; jc += CH) {
                double u[CH];
#pragma unroll
                for (int j = 0; j < CH / 2; ++j) {
                    const v2d w = sY[slot_of<PPL>(lane, jc / 2 + j)];
                    u[2 * j] = w.x;
                    u[2 * j + 1] = w.y;
                }
                if (full) {      // (wave-uniform)
#pragma unroll
                    for (int j = 0; j < CH; ++j) {
                        double r = u[j] - ka.hh;
#pragma unroll
                        for (int i = 0; i < D; ++i) r = fma(-ka.fw[i], z[i], r);
                        q = fma(r, r, q);
#pragma unroll
                        for (int i = 0; i < D; ++i) v[i] = fma(ka.wj[jc + j][i], r, v[i]);
                        double nz[D];
#pragma unroll
                        for (int i = 0; i < D; ++i) nz[i] = fma(ka.fd[i], z[i], fma(ka.fo[i], z[partner<D>(i)], fma(ka.fb[i], u[j], ka.fc[i])));
#pragma unroll
                        for (int i = 0; i < D; ++i) z[i] = nz[i];
                    }
                } else {         // the series' last tile: innovations of steps that do not exist do not count
#pragma unroll
                    for (int j = 0; j < CH; ++j) {
                        double r = u[j] - ka.hh;
#pragma unroll
                        for (int i = 0; i < D; ++i) r = fma(-ka.fw[i], z[i], r);
                        r = jc + j < nvalid ? r : 0.0;
                        q = fma(r, r, q);
#pragma unroll
                        for (int i = 0; i < D; ++i) v[i] = fma(ka.wj[jc + j][i], r, v[i]);
                        double nz[D];
#pragma unroll
                        for (int i = 0; i < D; ++i) nz[i] = fma(ka.fd[i], z[i], fma(ka.fo[i], z[partner<D>(i)], fma(ka.fb[i], u[j], ka.fc[i])));
#pragma unroll
                        for (int i = 0; i < D; ++i) z[i] = nz[i];
                    }
                }
            }
            // the state entering the tile goes in behind lane 0's steps: e_0 += M^N zin
            {
                double add[D];
#pragma unroll
                for (int i = 0; i < D; ++i) add[i] = fma(ka.lvr[0][i], zin[i], ka.lvi[0][i] * zin[partner<D>(i)]);
#pragma unroll
                for (int i = 0; i < D; ++i) z[i] += (lane == 0) ? add[i] : 0.0;
            }
            // ---- inclusive scan over the lanes: four levels inside the rows of 16 (row_shr moves with M^(N 2^k)), then the rows' totals across
#define TGP_LML_ROW_LEVEL(K)                                                                                         \
    do {                                                                                                             \
        double g_[D];                                                                                                \
        _Pragma("unroll") for (int i = 0; i < D; ++i) g_[i] = dpp_mov<0x110 + (1 << (K))>(z[i]);                    \
        _Pragma("unroll") for (int i = 0; i < D; ++i) z[i] = fma(ka.lvr[K][i], g_[i], fma(ka.lvi[K][i], g_[partner<D>(i)], z[i])); \
    } while (0)
            TGP_LML_ROW_LEVEL(0);
            TGP_LML_ROW_LEVEL(1);
            TGP_LML_ROW_LEVEL(2);
            TGP_LML_ROW_LEVEL(3);
#undef TGP_LML_ROW_LEVEL
            {
                double g[D];
#pragma unroll
                for (int i = 0; i < D; ++i) g[i] = dpp_mov<0x142, 0xA>(z[i]);      // rows 1, 3: the total of the row below
#pragma unroll
                for (int i = 0; i < D; ++i) z[i] = fma(mpr[i], g[i], fma(mpi[i], g[partner<D>(i)], z[i]));
#pragma unroll
                for (int i = 0; i < D; ++i) g[i] = dpp_mov<0x143, 0xC>(z[i]);      // rows 2, 3: everything up to lane 31 ...
                double g3[D];
#pragma unroll
                for (int i = 0; i < D; ++i) g3[i] = fma(ka.lvr[4][i], g[i], ka.lvi[4][i] * g[partner<D>(i)]);      // ... which row 3 sees through M^(16 N) more
#pragma unroll
                for (int i = 0; i < D; ++i) g[i] = lane >= 48 ? g3[i] : g[i];
#pragma unroll
                for (int i = 0; i < D; ++i) z[i] = fma(mpr[i], g[i], fma(mpi[i], g[partner<D>(i)], z[i]));
            }
            // the tile's true end state: the inclusive value of its last lane with a step in it
            const int le = full ? 63 : (int)((nv - 1) / N);
            double zend[D];
#pragma unroll
            for (int i = 0; i < D; ++i) zend[i] = readlane_d(z[i], le);
            // the state in front of the lane's steps: its left neighbour's (lane 0: the tile's)
#pragma unroll
            for (int i = 0; i < D; ++i) {
                const double sh = dpp_mov<0x138>(z[i]);
                z[i] = lane == 0 ? zin[i] : sh;
            }
            {
                // what the start state makes of the lane's sum of squares: q - 2 st . v + st' W st, W the sum of w_j' w_j over the lane's steps
                double ws[D], lin = 0.0, quad = 0.0;
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    double t = 0.0;
#pragma unroll
                    for (int k = 0; k < D; ++k) t = fma(ka.WN[i][k], z[k], t);
                    ws[i] = t;
                    lin = fma(z[i], v[i], lin);
                    quad = fma(z[i], t, quad);
                }
                if (!full) {      // (the series' last tile: lanes with fewer than N steps sum over the steps they have)
                    double qp = 0.0;
#pragma unroll 1
                    for (int j = 0; j < N; ++j) {
                        double t = 0.0;
#pragma unroll
                        for (int i = 0; i < D; ++i) t = fma(ka.wj[j][i], z[i], t);
                        qp = j < nvalid ? fma(t, t, qp) : qp;
                    }
                    quad = nvalid == N ? quad : qp;
                    // (V of a partial FIRST tile -- a series shorter than one tile -- would need the partial lanes' own W st: ws is used below as it is
                    //  for whole lanes; the one partial lane's share is recomputed from its steps)
                    if (nvalid != N) {
#pragma unroll
                        for (int i = 0; i < D; ++i) ws[i] = 0.0;
#pragma unroll 1
                        for (int j = 0; j < N; ++j) {
                            double t = 0.0;
#pragma unroll
                            for (int i = 0; i < D; ++i) t = fma(ka.wj[j][i], z[i], t);
                            t = j < nvalid ? t : 0.0;
#pragma unroll
                            for (int i = 0; i < D; ++i) ws[i] = fma(ka.wj[j][i], t, ws[i]);
                        }
                    }
                }
                acc += fma(-2.0, lin, q) + quad;
                if (first_tile) {
                    // the run's own start state (zero for now) moves the tile's innovations through V = sum_l (v_l - W st_l) M^(N l) (st_l: from a
                    // zero tile start, which is what this tile ran with): a row vector times the block form, (x P)_j = x_j pr_j + x_partner pi_partner
                    double pr[D], pi[D], x[D], xm[D];
                    lane_power(lane, pr, pi);
#pragma unroll
                    for (int i = 0; i < D; ++i) x[i] = v[i] - ws[i];
#pragma unroll
                    for (int i = 0; i < D; ++i) xm[i] = fma(x[i], pr[i], partner<D>(i) != i ? x[partner<D>(i)] * pi[partner<D>(i)] : 0.0);
#pragma unroll
                    for (int i = 0; i < D; ++i) V[i] = wave_sum(xm[i]);
                    first_tile = false;
                }
            }
#pragma unroll
            for (int i = 0; i < D; ++i) zin[i] = zend[i];
            lds_sync();      // (the slice may be overwritten from here on)
        }
        Q = wave_sum(acc);
#pragma unroll
        for (int i = 0; i < D; ++i) E[i] = zin[i];
        if (stamp) ka.part[kStampOff + 8 * blockIdx.x + 2] = (double)wall_clock64();
    }
    if (head_flag_due && active) {      // (the stores were issued a whole run ago: nothing to wait for)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        raise_head_flag();
    }
    __builtin_amdgcn_s_setprio(0);
    if (lane == 0) {
        double* p = sPost + wave * PW;
        p[0] = Q;
        p[1] = active ? 1.0 : 0.0;
#pragma unroll
        for (int i = 0; i < D; ++i) {
            p[2 + i] = V[i];
            p[2 + D + i] = E[i];
        }
    }
    __syncthreads();
    if (stamp) ka.part[kStampOff + 8 * blockIdx.x + 3] = (double)wall_clock64();
    if (threadIdx.x == 0) {
        // the runs of this workgroup close each other: run w starts from the end state of run w - 1; run 0 is the host's to close
        double tot = sPost[0], el[D];
#pragma unroll
        for (int i = 0; i < D; ++i) el[i] = sPost[2 + D + i];
        for (int w = 1; w < kNW; ++w) {
            const double* p = sPost + w * PW;
            if (p[1] == 0.0) break;
            double lin = 0.0, quad = 0.0;
#pragma unroll
            for (int i = 0; i < D; ++i) {
                lin = fma(el[i], p[2 + i], lin);
                double v = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) v = fma(ka.Wt[i][k], el[k], v);
                quad = fma(el[i], v, quad);
            }
            tot += p[0] - 2.0 * lin + quad;
#pragma unroll
            for (int i = 0; i < D; ++i) el[i] = p[2 + D + i];
        }
        // the workgroup's record: 1 + 2 D (value, check) pairs, each ONE 16-byte write-through store -- check = the value's bits ^ the call's key, so
        // the host knows a pair of THIS call when it sees one and needs no stream synchronisation to read the records (tgp_lml.hpp await_records)
        v2d* out = reinterpret_cast<v2d*>(ka.part) + (size_t)blockIdx.x * (1 + 2 * D);
        const long long key = (long long)record_key(ka.seq);
        auto put = [&](int k, double v) {
            v2d rec;
            rec.x = v;
            rec.y = __longlong_as_double(__double_as_longlong(v) ^ key);
            v2d* dst = out + k;
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(rec) : "memory");
        };
        put(0, tot);
#pragma unroll
        for (int i = 0; i < D; ++i) {
            put(1 + i, sPost[2 + i]);
            put(1 + D + i, el[i]);
        }
        if (stamp) ka.part[kStampOff + 8 * blockIdx.x + 4] = (double)wall_clock64();
    }
}

#undef ka
// ---- host: block-form helpers (the plan's convention: (P x)_i = pr_i x_i + pi_i x_partner; partner = i ^ 1 inside the 2 npair leading components)
inline int hpartner(int i, int np) { return i < 2 * np ? (i ^ 1) : i; }
void bsq(int d, double* pr, double* pi) {      // (re, im) <- (re, im)^2; the sign convention of im is preserved
    for (int i = 0; i < d; ++i) {
        const double r = pr[i] * pr[i] - pi[i] * pi[i], im = 2.0 * pr[i] * pi[i];
        pr[i] = r;
        pi[i] = im;
    }
}
void bmulv(int d, int np, const double* pr, const double* pi, const double* x, double* out) {      // out = P x
    for (int i = 0; i < d; ++i) {
        const int p = hpartner(i, np);
        out[i] = pr[i] * x[i] + (p != i ? pi[i] * x[p] : 0.0);
    }
}
void vmulb(int d, int np, const double* pr, const double* pi, const double* x, double* out) {      // out = x P (a row vector)
    for (int j = 0; j < d; ++j) {
        const int p = hpartner(j, np);
        out[j] = x[j] * pr[j] + (p != j ? x[p] * pi[p] : 0.0);
    }
}

template <int D, int N>
void fill(LArgs<D, N>& ka, const tgp_plan::Modal& md) {
    for (int i = 0; i < D; ++i) {
        ka.fd[i] = md.fd[i];
        ka.fo[i] = md.fo[i];
        ka.fb[i] = md.fb[i];
        ka.fc[i] = md.fa[i] - md.fb[i] * md.hh;
        ka.fw[i] = md.fw[i];
    }
    ka.hh = md.hh;
    // M^(N 2^k)
    double pr[tgp_plan::kMaxD], pi[tgp_plan::kMaxD];
    for (int i = 0; i < D; ++i) {
        pr[i] = md.fd[i];
        pi[i] = md.fo[i];
    }
    for (int s = 1; s < N; s <<= 1) bsq(D, pr, pi);
    for (int k = 0; k < 6; ++k) {
        for (int i = 0; i < D; ++i) {
            ka.lvr[k][i] = pr[i];
            ka.lvi[k][i] = pi[i];
        }
        bsq(D, pr, pi);
    }
    // wj[j] = fw' M^j; WN = sum_j wj[j]' wj[j]
    {
        const int np = md.npair;
        double x[tgp_plan::kMaxD], nx[tgp_plan::kMaxD];
        for (int i = 0; i < D; ++i) {
            x[i] = md.fw[i];
            for (int k = 0; k < D; ++k) ka.WN[i][k] = 0.0;
        }
        for (int j = 0; j < N; ++j) {
            for (int i = 0; i < D; ++i) {
                ka.wj[j][i] = x[i];
                for (int k = 0; k < D; ++k) ka.WN[i][k] += x[i] * x[k];
            }
            vmulb(D, np, md.fd, md.fo, x, nx);
            for (int i = 0; i < D; ++i) x[i] = nx[i];
        }
    }
}

template <int D, int N>
int launch(hipStream_t st, const tgp_plan::Modal& md, const Geometry& g, long long T, const double* y, const Buffers& b, long long seq, const double* Wt) {
    static_assert(sizeof(LArgs<D, N>) <= 8192, "the kernel-argument segment");
    LArgs<D, N> a;
    fill<D, N>(a, md);
    for (int i = 0; i < D; ++i)
        for (int k = 0; k < D; ++k) a.Wt[i][k] = Wt[i * D + k];
    a.T = T;
    a.nhs = md.nhs;
    a.G = g.G;
    a.W = g.W;
    a.Chi = g.Chi;
    a.Clo = g.Clo;
    a.y = y;
    a.part = b.part;
    a.head_in = b.head_in;
    a.flags = b.flags;
    a.seq = seq;
    a.nwg = g.nwg;
    {
        static const int dbg = [] {
            const char* v = std::getenv("TGP_LML_DBG");
            return v ? std::atoi(v) : 0;
        }();
        a.dbg = dbg;
    }
    static const size_t lds_pad = [] {      // TGP_LML_LDS_PAD=<bytes>: development (does a launch that asks for more LDS start later?)
        const char* v = std::getenv("TGP_LML_LDS_PAD");
        return v ? (size_t)std::atol(v) : (size_t)0;
    }();
    const size_t lds = (size_t)kNW * 64 * (N / 2) * 16 + (size_t)kNW * (2 + 2 * D) * sizeof(double) + lds_pad;
    const bool aligned = (reinterpret_cast<uintptr_t>(y) & 15) == 0;
    // (the limit on dynamic LDS is a per-device attribute of the function)
    static bool attr_done_dev[64][2] = {{false, false}};
    int dev = 0;
    (void)hipGetDevice(&dev);
    bool& attr_done = attr_done_dev[dev & 63][aligned ? 1 : 0];
    if (!attr_done) {
        const void* fn = aligned ? reinterpret_cast<const void*>(&k_lml_stream<D, N, true>) : reinterpret_cast<const void*>(&k_lml_stream<D, N, false>);
        const hipError_t rc = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (rc != hipSuccess) return (int)rc;
        attr_done = true;
    }
    if (aligned) hipLaunchKernelGGL((k_lml_stream<D, N, true>), dim3(g.nwg), dim3(kNW * 64), lds, st, a);
    else hipLaunchKernelGGL((k_lml_stream<D, N, false>), dim3(g.nwg), dim3(kNW * 64), lds, st, a);
    return (int)hipGetLastError();
}

}  // namespace

Geometry choose_geometry(const tgp_plan::Modal& md, long long T) {
    Geometry g;
    static const int forced = [] {
        const char* v = std::getenv("TGP_LML_N");
        const int n = v ? std::atoi(v) : 0;
        return (n == 16 || n == 32) ? n : 0;
    }();
    // sixteen steps per lane and four waves per SIMD where the registers allow it (d <= 2), thirty-two and two beyond
    g.n = forced ? forced : (md.d <= 2 ? 16 : 32);      // (d = 3 at sixteen steps per lane: 28 registers spilled under the four-waves budget)
    if (64 * g.n < md.halo) g.n = 32;      // (a tile must outlast the halo: 2048 >= kHaloMax)
    const long long Tp = T - md.nhs, tile = 64LL * g.n;
    g.G = (Tp + tile - 1) / tile;          // tiles behind the head; the last one may be partial
    g.W = Tp / tile;                       // whole tiles: a run holds at least one (a partial last tile is its run's second or later)
    // tiles per SIMD: the fewest (Chi + Clo) that the workgroups the chip holds at once cover -- four (n = 16) or two waves per SIMD: what LDS and registers
    // hold -- and two at least (every wave of a workgroup a run: the head's hand-over rides on the last one) once there are tiles for that
    const long long wgmax = g.n == 16 ? kMaxWG : kMaxWG / 2;
    long long S = (g.W + 4 * wgmax - 1) / (4 * wgmax);
    if (S < 2) S = 2;
    g.Chi = (S + 1) / 2;
    g.Clo = S / 2;
    g.nwg = (int)std::max<long long>(1, (g.W + 4 * S - 1) / (4 * S));
    g.R = 0;      // (runs that hold a tile: what the tests and the debug print report)
    for (long long b = 0; b < g.nwg; ++b)
        for (int w = 0; w < kNW; ++w) {
            const long long g0 = b * 4 * S + (w < 4 ? w * g.Chi : 4 * g.Chi + (w - 4) * g.Clo);
            g.R += (g.W > 0 ? g0 < g.W : (b == 0 && w == 0)) ? 1 : 0;
        }
    g.C = g.Chi;
    g.first_tile = Tp < tile ? Tp : tile;
    return g;
}

// Wt = sum_{t < n} w_t' w_t with w_t = fw' M^t, by the bits of n: W(2 m) = W(m) + (M^m)' W(m) M^m, W(m + 1) = W(m) + w_m' w_m
void quad_table(const tgp_plan::Modal& md, long long n, double* Wt) {
    const int d = md.d, np = md.npair;
    constexpr int MD = tgp_plan::kMaxD;
    double W[MD][MD] = {{0.0}}, pr[MD], pi[MD];      // W(m), M^m
    for (int i = 0; i < d; ++i) {
        pr[i] = 1.0;
        pi[i] = 0.0;
    }
    int top = 0;
    while ((n >> top) > 1) ++top;
    if (n <= 0) top = -1;
    for (int bit = top; bit >= 0; --bit) {
        // double: W <- W + P' W P
        double X[MD][MD], Y[MD][MD];
        for (int i = 0; i < d; ++i) vmulb(d, np, pr, pi, W[i], X[i]);      // rows of W times P
        for (int j = 0; j < d; ++j) {                                        // columns: Y = P' X, (P' X)_ij = sum_k P_ki X_kj = pr_i X_ij + pi_partner X_partner,j
            for (int i = 0; i < d; ++i) {
                const int p = hpartner(i, np);
                Y[i][j] = pr[i] * X[i][j] + (p != i ? pi[p] * X[p][j] : 0.0);
            }
        }
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j) W[i][j] += Y[i][j];
        bsq(d, pr, pi);
        if ((n >> bit) & 1) {
            double w[MD];
            vmulb(d, np, pr, pi, md.fw, w);      // w_m = fw' M^m
            for (int i = 0; i < d; ++i)
                for (int j = 0; j < d; ++j) W[i][j] += w[i] * w[j];
            // P <- P M
            for (int i = 0; i < d; ++i) {
                const double r = pr[i] * md.fd[i] - pi[i] * md.fo[i], im = pr[i] * md.fo[i] + pi[i] * md.fd[i];
                pr[i] = r;
                pi[i] = im;
            }
        }
    }
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) Wt[i * d + j] = 0.5 * (W[i][j] + W[j][i]);
}

int enqueue(hipStream_t stream, const tgp_plan::Modal& md, const Geometry& g, long long T, const double* y, const Buffers& b, long long seq, const double* Wt,
            const char** kname) {
    if (kname) *kname = g.n == 32 ? "k_lml_stream<32>" : "k_lml_stream<16>";
#define TGP_LML_CASE(DD)                                                                                      \
    case DD:                                                                                                  \
        return g.n == 32 ? launch<DD, 32>(stream, md, g, T, y, b, seq, Wt) : launch<DD, 16>(stream, md, g, T, y, b, seq, Wt);
    switch (md.d) {
        TGP_LML_CASE(1)
        TGP_LML_CASE(2)
        TGP_LML_CASE(3)
        TGP_LML_CASE(4)
        TGP_LML_CASE(5)
        TGP_LML_CASE(6)
        TGP_LML_CASE(7)
        TGP_LML_CASE(8)
    }
#undef TGP_LML_CASE
    return (int)hipErrorInvalidValue;
}

double finish(const tgp_plan::Modal& md, const Geometry& g, const double* part, const double* z0, const double* Wt) {
    const int d = md.d, pw = 1 + 2 * d;
    double s = 0.0;
    double stv[tgp_plan::kMaxD];
    for (int i = 0; i < d; ++i) stv[i] = z0[i];
    for (int w = 0; w < g.nwg; ++w) {
        const double* p = part + (size_t)w * pw * 2;      // (value, check) pairs
        double lin = 0.0, quad = 0.0;
        for (int i = 0; i < d; ++i) {
            lin += stv[i] * p[2 * (1 + i)];
            double v = 0.0;
            for (int k = 0; k < d; ++k) v += Wt[i * d + k] * stv[k];
            quad += stv[i] * v;
        }
        s += p[0] - 2.0 * lin + quad;
        for (int i = 0; i < d; ++i) stv[i] = p[2 * (1 + d + i)];
    }
    return s;
}

bool records_there(const Geometry& g, int d, const double* part, long long seq, size_t* next) {
    const size_t n = (size_t)g.nwg * (1 + 2 * d);
    const unsigned long long key = record_key(seq);
    const volatile unsigned long long* q = reinterpret_cast<const volatile unsigned long long*>(part);
    size_t k = *next;
    for (; k < n; ++k) {
        const unsigned long long v = q[2 * k], c = q[2 * k + 1];
        if ((v ^ c) != key) break;
    }
    *next = k;
    if (k == n) __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return k == n;
}

}  // namespace tgp_lml
