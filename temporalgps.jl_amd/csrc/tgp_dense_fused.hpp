// Dense engine, mid-sized states (16 < d <= 64, p <= 16): the whole recursion in ONE persistent kernel per pass.
//
// At these sizes a time step is ~1e5 flop: the seven-launch chain of tgp_dense.hip costs ~26 us per step in dependent
// dispatches alone (measured, d = 17..64), three times what a NumPy loop on the host needs. Here one 256-thread workgroup walks
// the time steps itself: P and A P live in LDS, the d x d products run on v_mfma_f64_16x16x4_f64 (one 16 x 16 tile per wave and
// pass), A's MFMA fragments and the Q tiles stay in registers when the model shares them across steps, and the p observations
// of a step are applied as p scalar Kalman updates (lgc.jl:247-257 -- algebraically the joint update of lgc.jl:129-141 for
// diagonal noise, and what the scan engine does for vector observations). No launch per step, ~0.5 us per step.
//
// Included by tgp_dense.hip (namespace tgp_dense, after the MFMA helpers).
#pragma once

struct FusedArgs {
    int64_t T = 0;
    int64_t step0 = 0, step1 = 0;      // processing steps [step0, step1) of this launch (long series are cut into several launches)
    int d = 0, p = 0, Pq = 0, ordering = 0;
    const double *A = nullptr, *Q = nullptr, *H = nullptr, *a = nullptr, *h = nullptr, *R = nullptr;   // padded blocks (tgp_dense.hip layouts)
    int64_t sA = 0, sQ = 0, sH = 0, sa = 0, sh = 0, sR = 0;                                         // strides per time step (0 = shared)
    const double* x0 = nullptr;        // padded P (DP x DP, column-major) followed by m (DP)
    const double* y = nullptr;         // [T][p]
    const uint8_t* mask = nullptr;     // [T][p] or null
    double* m_out = nullptr;           // [T][d] filtering means (nullable)
    double* P_out = nullptr;           // [T][d*d] filtering covariances, column-major (nullable)
    double* result8 = nullptr;         // [0] += lml, [1] += missing count, [2] = first step with a non-positive innovation variance + 1
    double* xfin = nullptr;            // final state, same layout as x0 (nullable)
    double* marg_mean = nullptr;       // prior marginals instead of filtering (marginals(model), lgssm.jl:99-115): [T][p] mean and
    double* marg_var = nullptr;        // variance of every emission; y / mask are not read, the state is only predicted
    double* aux_out = nullptr;         // [T][p][d + 2]: per scalar update v = P h' (d), s = h v + R, nu = y - h m - hh: what the
                                       // backward pass (dk_fused_smooth) needs of the filter (nullable)
};

// sum over the 64 lanes: four row_shr DPP steps inside each row of 16 lanes (VALU, no LDS crossbar), then the four row totals
// through v_readlane
template <int CTRL>
__device__ inline double dpp_shr(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ inline double wave_sum(double v) {
    v += dpp_shr<0x111>(v);      // row_shr:1
    v += dpp_shr<0x112>(v);      // row_shr:2
    v += dpp_shr<0x114>(v);      // row_shr:4
    v += dpp_shr<0x118>(v);      // row_shr:8   -> lane 15 of every row holds the row's sum
    return (rdlane(v, 15) + rdlane(v, 31)) + (rdlane(v, 47) + rdlane(v, 63));
}

template <int DP>
struct FusedCfg {
    static constexpr int NT = DP / 16, KS = DP / 4, LD = DP + 4;
    static constexpr int NSL = 4 * NT;                          // partial sums per element of a matrix-vector product (row slices)
    static constexpr int HPT = (16 * DP) / 256;                 // emission-row elements per thread (per-step H prefetch)
    // LDS (doubles): P | T1' | m | a | v | partial sums (v) | partial sums (A m) | H rows | scalars of the step
    static constexpr int oP = 0, oT = oP + DP * LD, oM = oT + DP * LD, oA = oM + DP, oV = oA + DP, oRed = oV + DP, oRedM = oRed + NSL * DP,
                         oH = oRedM + 4 * DP, oS = oH + 16 * DP, TOTAL = oS + 64;
    static constexpr size_t LDS_BYTES = (size_t)TOTAL * sizeof(double);
};

// The covariance lives in REGISTERS between the products. Wave w belongs to block index b = w % NT: in A P it computes tiles
// (b, x) -- A operand = the fragments of A's block row b -- and in (A P)A' + Q the tiles (x, b) -- B operand = the SAME
// fragments -- so a wave keeps one block row of A (KS doubles per lane) instead of all of A. The tiles (x, b) stay in the MFMA
// accumulator layout (lane (lr, lq), register r <-> row 16 x + lq + 4 r, column 16 b + lr): v = h P is formed from them (each
// lane: 4 rows of one column; the 4 NT row slices are summed by wave 0), the rank-1 downdate is 4 FMAs per tile and lane, and only
// then a tile goes to LDS, where the next step's A P reads it as the B operand.
template <int DP>
__global__ __launch_bounds__(256) void dk_fused_filter(const FusedArgs g) {
    using C = FusedCfg<DP>;
    constexpr int NT = C::NT, KS = C::KS, LD = C::LD, NSL = C::NSL, HPT = C::HPT;
    constexpr int NGW = 4 / NT > 0 ? 4 / NT : 1;       // waves per block index (NT = 2: two waves share a block row, one x each)
    constexpr int XPW = (NT + NGW - 1) / NGW;          // tiles per wave
    extern __shared__ double lds[];
    double* sP = lds + C::oP;
    double* sT = lds + C::oT;
    double* sm = lds + C::oM;
    double* sa = lds + C::oA;
    double* sv = lds + C::oV;
    double* red = lds + C::oRed;       // [NSL][DP] partial sums of v = h P
    double* redm = lds + C::oRedM;     // [4][DP] partial sums of A m
    double* sH = lds + C::oH;
    double* ss = lds + C::oS;          // [0] 1/s of the current scalar update; [16..31] h; [32..47] R
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const int b = w % NT, grp = w / NT;
    const bool mfma_wave = w < NT * NGW;               // (NT = 3: wave 3 has no tiles)
    const bool fwd = g.ordering == 0;
    const bool marg = g.marg_mean != nullptr;
    const bool H_shared = g.sH == 0 && g.sh == 0 && g.sR == 0;
    auto tstep = [&](int64_t step) { return fwd ? step : g.T - 1 - step; };

    // ---- state and the shared model blocks
    for (int e = tid; e < DP * DP; e += 256) sP[(e % DP) * LD + e / DP] = g.x0[e];      // sP[i][j] = P[i][j] (row i)
    if (tid < DP) sm[tid] = g.x0[DP * DP + tid];
    double af[KS];                // A fragments of block row b: af[ks] = A[16 b + lr][4 ks + lq]
    double qf[XPW][4];            // Q tiles (x, b) of this wave
    d4 pt[XPW];                   // the covariance tiles (x, b) themselves
    auto load_A = [&](const double* A) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) af[ks] = A[(b * 16 + lr) + (int64_t)(ks * 4 + lq) * DP];
    };
    auto load_Q = [&](const double* Q) __attribute__((always_inline)) {
#pragma unroll
        for (int x = 0; x < NT; ++x)
            if (x % NGW == grp) {
#pragma unroll
                for (int r = 0; r < 4; ++r) qf[x / NGW][r] = Q[(x * 16 + lq + 4 * r) + (int64_t)(b * 16 + lr) * DP];
            }
    };
    // ---- per-step inputs travel one step ahead in registers: issued at the top of step s for step s + 1, so that their latency
    //      hides behind a whole step (barriers below only wait for LDS traffic: lds_barrier)
    double a_n = 0.0;                 // a[tid] (tid < DP)
    double y_n = 0.0;                 // lane j < p of wave 0: y[t][j], and its missing flag
    int miss_n = 0;
    double hrow_n[HPT];               // emission rows (per-step H only)
    double hs_n = 0.0, rs_n = 0.0;    // h[j], R[j] for tid = j < p (per-step emissions only)
    auto fetch = [&](int64_t t) __attribute__((always_inline)) {
        if (tid < DP) a_n = g.a[t * g.sa + tid];
        if (tid < g.p && !marg) {
            miss_n = g.mask != nullptr && g.mask[t * g.p + tid] != 0;
            y_n = g.y[t * g.p + tid];
        }
        if (!H_shared) {
            const double* H = g.H + t * g.sH;
#pragma unroll
            for (int u = 0; u < HPT; ++u) {
                const int e = tid + u * 256, j = e / DP, k = e % DP;
                hrow_n[u] = j < g.p ? H[j + (int64_t)k * g.Pq] : 0.0;
            }
            if (tid < g.p) {
                hs_n = g.h[t * g.sh + tid];
                rs_n = g.R[t * g.sR + tid];
            }
        }
    };
    auto stage_H = [&]() __attribute__((always_inline)) {      // prefetched emission block -> LDS (row j: sH[j DP + k])
#pragma unroll
        for (int u = 0; u < HPT; ++u) sH[tid + u * 256] = hrow_n[u];
        if (tid < g.p) {
            ss[16 + tid] = hs_n;
            ss[32 + tid] = rs_n;
        }
    };
    if (g.sA == 0) load_A(g.A);
    if (g.sQ == 0) load_Q(g.Q);
    if (H_shared) {
        for (int e = tid; e < g.p * DP; e += 256) sH[e] = g.H[(e / DP) + (int64_t)(e % DP) * g.Pq];
        if (tid < g.p) {
            ss[16 + tid] = g.h[tid];
            ss[32 + tid] = g.R[tid];
        }
    }
    fetch(tstep(g.step0));
    if (!H_shared) stage_H();
    double lml = 0.0, nmiss = 0.0, bad = 0.0, sprod = 1.0;      // (wave 0, lane 0) log det via a running product, as the scan kernels do
    __syncthreads();
    auto tiles_from_lds = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int x = 0; x < NT; ++x)
            if (x % NGW == grp && mfma_wave) {
#pragma unroll
                for (int r = 0; r < 4; ++r) pt[x / NGW][r] = sP[(x * 16 + lq + 4 * r) * LD + b * 16 + lr];
            }
    };
    auto tiles_to_lds = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int x = 0; x < NT; ++x)
            if (x % NGW == grp && mfma_wave) {
#pragma unroll
                for (int r = 0; r < 4; ++r) sP[(x * 16 + lq + 4 * r) * LD + b * 16 + lr] = pt[x / NGW][r];
            }
    };
    tiles_from_lds();

    for (int64_t step = g.step0; step < g.step1; ++step) {
        const int64_t t = tstep(step);
        // this step's prefetched inputs; then the loads of the next step
        const double y_c = y_n;
        const int miss_c = miss_n;
        if (tid < DP) sa[tid] = a_n;          // read after the barrier inside predict
        if (step + 1 < g.step1) fetch(tstep(step + 1));

        // predict (lgc.jl:46-52): (sm, sP) hold the state; afterwards sm holds A m + a and the tiles hold A P A' + Q (registers only)
        auto predict = [&]() __attribute__((always_inline)) {
            if (g.sA != 0) load_A(g.A + t * g.sA);
            if (g.sQ != 0) load_Q(g.Q + t * g.sQ);
            // T1 = A P: tiles (b, x), stored TRANSPOSED (sT[col][row])
#pragma unroll
            for (int x = 0; x < NT; ++x)
                if (x % NGW == grp && mfma_wave) {
                    d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) acc = mfma_f64(af[ks], sP[(ks * 4 + lq) * LD + x * 16 + lr], acc);
#pragma unroll
                    for (int r = 0; r < 4; ++r) sT[(x * 16 + lr) * LD + b * 16 + lq + 4 * r] = acc[r];
                }
            // A m on the vector pipe while the matrix pipe works: block row b by its first wave, lane (lr, lq) sums its k = 4 ks + lq
            if (grp == 0 && mfma_wave) {
                double sacc = 0.0;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) sacc = fma(af[ks], sm[ks * 4 + lq], sacc);
                redm[lq * DP + b * 16 + lr] = sacc;
            }
            lds_barrier();
            // Pp = T1 A' + Q: tiles (x, b) -> registers
#pragma unroll
            for (int x = 0; x < NT; ++x)
                if (x % NGW == grp && mfma_wave) {
                    d4 acc = d4{qf[x / NGW][0], qf[x / NGW][1], qf[x / NGW][2], qf[x / NGW][3]};
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) acc = mfma_f64(sT[(ks * 4 + lq) * LD + x * 16 + lr], af[ks], acc);
                    pt[x / NGW] = acc;
                }
            if (tid < DP) sm[tid] = ((redm[tid] + redm[DP + tid]) + (redm[2 * DP + tid] + redm[3 * DP + tid])) + sa[tid];
        };

        // partial sums of v = h_j P from the register tiles (lgc.jl:249: V = h'P, i.e. column sums): lane (lr, lq) of tile (x, b)
        // adds its 4 rows of column 16 b + lr; slice (x, lq) of the NSL = 4 NT partial sums
        auto v_partials = [&](int j) __attribute__((always_inline)) {
            const double* hj = sH + j * DP;
#pragma unroll
            for (int x = 0; x < NT; ++x)
                if (x % NGW == grp && mfma_wave) {
                    double sacc = 0.0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) sacc = fma(pt[x / NGW][r], hj[x * 16 + lq + 4 * r], sacc);
                    red[(x * 4 + lq) * DP + b * 16 + lr] = sacc;
                }
        };
        // the p scalar updates of step t (lgc.jl:247-257 each); lml, missing count, not-PD flag accumulate in wave 0
        auto update = [&]() __attribute__((always_inline)) {
            for (int j = 0; j < g.p; ++j) {
                v_partials(j);
                lds_barrier();
                if (w == 0) {
                    bool continue_marg = false;
                    double v = 0.0, hi = 0.0, mi = 0.0;
                    if (lane < DP) {
#pragma unroll
                        for (int q = 0; q < NSL; ++q) v += red[q * DP + lane];
                        hi = sH[j * DP + lane];
                        mi = sm[lane];
                    }
                    if (marg) {          // emission marginal of the (predicted) state: N(h m + hh, h P h' + R); the state does not move
                        const double var = wave_sum(hi * v) + ss[32 + j], mean = wave_sum(hi * mi) + ss[16 + j];
                        if (lane == 0) {
                            g.marg_mean[t * g.p + j] = mean;
                            g.marg_var[t * g.p + j] = var;
                            ss[0] = 0.0;
                        }
                        if (lane < DP) sv[lane] = 0.0;
                        continue_marg = true;
                    }
                    const bool miss = !marg && __builtin_amdgcn_readlane(miss_c, j) != 0;
                    const double s = marg ? 1.0 : wave_sum(hi * v) + (miss ? kLargeVar : ss[32 + j]);
                    const double nu = marg ? 0.0 : (miss ? 0.0 : rdlane(y_c, j)) - wave_sum(hi * mi) - ss[16 + j];
                    const double sinv = 1.0 / s;
                    if (lane < DP && !continue_marg) {
                        sv[lane] = v;
                        sm[lane] = mi + v * sinv * nu;
                    }
                    if (g.aux_out && !continue_marg) {
                        double* ax = g.aux_out + (t * g.p + j) * (int64_t)(g.d + 2);
                        if (lane < g.d) ax[lane] = v;
                        if (lane == 0) {
                            ax[g.d] = s;
                            ax[g.d + 1] = nu;
                        }
                    }
                    if (lane == 0 && !continue_marg) {
                        ss[0] = sinv;
                        lml += -0.5 * (kLog2Pi + nu * nu * sinv) + (miss ? 0.5 * (kLog2Pi + log(kLargeVar)) : 0.0);
                        sprod *= s;
                        if (sprod > 1e100 || sprod < 1e-100) {
                            lml -= 0.5 * log(sprod);
                            sprod = 1.0;
                        }
                        nmiss += miss ? 1.0 : 0.0;
                        if (!(s > 0.0) && bad == 0.0) bad = (double)(t + 1);
                    }
                }
                lds_barrier();
                {   // P -= v v' / s in the register tiles
                    const double vc = sv[b * 16 + lr] * ss[0];
#pragma unroll
                    for (int x = 0; x < NT; ++x)
                        if (x % NGW == grp && mfma_wave) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) pt[x / NGW][r] = fma(-sv[x * 16 + lq + 4 * r], vc, pt[x / NGW][r]);
                        }
                }
                // the emission block of the NEXT step may replace this one once its last reader (wave 0 above) is done
                if (!H_shared && j == g.p - 1 && step + 1 < g.step1) stage_H();
                if (j + 1 < g.p) lds_barrier();      // (sv, ss[0] and red are rewritten by the next scalar update)
            }
            tiles_to_lds();
            lds_barrier();
        };
        auto emit = [&]() __attribute__((always_inline)) {       // filtering distribution of step t (sP, sm are current)
            if (g.m_out && tid < g.d) g.m_out[t * g.d + tid] = sm[tid];
            if (g.P_out) {
                double* Po = g.P_out + t * (int64_t)g.d * g.d;
                for (int e = tid; e < g.d * g.d; e += 256) Po[e] = sP[(e % g.d) * LD + e / g.d];
            }
        };
#ifdef FUSED_SKIP      // development: time the phases in isolation (scripts/fused_kbench.hip)
        if (!(FUSED_SKIP & 1)) { predict(); lds_barrier(); }
        if (!(FUSED_SKIP & 2)) update();
        emit();
#else
        if (fwd) {
            predict();
            lds_barrier();        // sm complete (A m + a) before wave 0 reads it
            update();
            emit();
        } else {      // Reverse (lgssm.jl:161-165, 183-187): update the carried state, then predict with the same step's transition
            update();
            emit();
            lds_barrier();        // everybody is done with sP / sm of the filtering state before predict's results replace them
            predict();
            lds_barrier();
        }
#endif
    }
    __syncthreads();
    if (!fwd) {                   // Reverse: the carried (predicted) covariance is in the register tiles only
        tiles_to_lds();
        __syncthreads();
    }
    if (g.xfin) {
        for (int e = tid; e < DP * DP; e += 256) g.xfin[e] = sP[(e % DP) * LD + e / DP];
        if (tid < DP) g.xfin[DP * DP + tid] = sm[tid];
    }
    if (tid == 0) {
        g.result8[0] += lml - 0.5 * log(sprod);
        g.result8[1] += nmiss;
        if (bad != 0.0 && g.result8[2] == 0.0) g.result8[2] = bad;
    }
}

// ------------------------------------------------------------------------------------------------ rand (shared A, a, Q)
// rand(rng, model) with supplied noise for mid-sized states (lgssm.jl:65-91): x <- A x + a + Lq eps_t with Lq = chol(Q + 1e-9 I)
// (lower factor, computed beforehand by the blocked factorisation), y = H x + h + sqrt(R [+ 1e-9]) .* eps_e. One workgroup
// walks the steps; A' and Lq' live in LDS so that thread (i, slice) reads consecutive addresses.
struct FusedRandArgs {
    int64_t T = 0;
    int d = 0, p = 0, Pq = 0, ordering = 0, small_out = 0;
    const double *A = nullptr, *Lq = nullptr, *a = nullptr, *H = nullptr, *h = nullptr, *R = nullptr;     // padded; A, a, Lq shared
    int64_t sH = 0, sh = 0, sR = 0;
    const double* x0 = nullptr;        // [DP] drawn initial state (padded with zeros)
    const double* eps_t = nullptr;     // [T][d]
    const double* eps_e = nullptr;     // [T][p]
    double* y_out = nullptr;           // [T][p]
};
template <int DP>
struct FusedRandCfg {
    static constexpr int LD = DP + 1, NG = 256 / DP;
    static constexpr int oA = 0, oL = oA + DP * LD, oX = oL + DP * LD, oE = oX + DP, oRed = oE + DP, TOTAL = oRed + NG * DP;
    static constexpr size_t LDS_BYTES = (size_t)TOTAL * sizeof(double);
};
template <int DP>
__global__ __launch_bounds__(256) void dk_fused_rand(const FusedRandArgs g) {
    using C = FusedRandCfg<DP>;
    constexpr int LD = C::LD, NG = C::NG;
    extern __shared__ double lds[];
    double* sAt = lds + C::oA;      // sAt[k LD + i] = A[i][k]
    double* sLt = lds + C::oL;      // sLt[k LD + i] = Lq[i][k]
    double* sx = lds + C::oX;
    double* se = lds + C::oE;
    double* red = lds + C::oRed;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int e = tid; e < DP * DP; e += 256) {
        const int i = e % DP, k = e / DP;
        sAt[k * LD + i] = g.A[e];                                 // column-major A: element e = A[i][k]
        sLt[k * LD + i] = i >= k ? g.Lq[e] : 0.0;                 // lower factor (entries above the diagonal are not part of it)
    }
    if (tid < DP) sx[tid] = g.x0[tid];
    __syncthreads();
    const bool fwd = g.ordering == 0;
    auto emit = [&](int64_t t) __attribute__((always_inline)) {     // y[t][j] = h_j x + hh_j + sqrt(R_j) eps_e[t][j]: wave j % 4, lanes over k
        for (int j = w; j < g.p; j += 4) {
            double acc = 0.0;
            if (lane < DP) acc = g.H[t * g.sH + j + (int64_t)lane * g.Pq] * sx[lane];
            const double hx = wave_sum(acc);
            if (lane == 0) {
                const double Rv = g.R[t * g.sR + j];
                g.y_out[t * g.p + j] = hx + g.h[t * g.sh + j] + sqrt(g.small_out ? Rv + 1e-9 : Rv) * g.eps_e[t * g.p + j];
            }
        }
    };
    auto move = [&](int64_t t) __attribute__((always_inline)) {     // x <- A x + a + Lq eps_t[trans index]
        if (tid < DP) se[tid] = tid < g.d ? g.eps_t[t * g.d + tid] : 0.0;
        lds_barrier();
        const int i = tid % DP, sl = tid / DP;
        if (sl < NG) {
            double s = 0.0;
            for (int k = sl; k < DP; k += NG) s += sAt[k * LD + i] * sx[k] + sLt[k * LD + i] * se[k];
            red[sl * DP + i] = s;
        }
        lds_barrier();
        if (tid < DP) {
            double v = g.a[tid];
#pragma unroll
            for (int q = 0; q < NG; ++q) v += red[q * DP + tid];
            sx[tid] = v;
        }
        lds_barrier();
    };
    for (int64_t step = 0; step < g.T; ++step) {
        const int64_t t = fwd ? step : g.T - 1 - step;
        if (fwd) {
            move(t);
            emit(t);
        } else {
            emit(t);
            lds_barrier();
            move(t);
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward pass
// marginals(replace_observation_noise_cov(posterior(model, y), Rnew)) for mid-sized states WITHOUT the d x d Cholesky of the
// RTS form (lgssm.jl:231-238): the modified Bryson-Frazier recursion carries the adjoint pair (lambda, Lambda) backwards,
//     per scalar update (reverse order):  K = v / s,  q = Lambda K,
//         Lambda <- Lambda - h' q' - q h + (K'q + 1/s) h'h,      lambda <- lambda - h' (K'lambda + nu / s)
//     per transition:                      Lambda <- A' Lambda A,  lambda <- A' lambda          (two MFMA products)
//     smoothed state of step t:            m_s = m_f - P_f lambda,  P_s = P_f - P_f Lambda P_f   (filtering state from the forward pass)
// and the emission marginals need only w = P_f h':  mean = h m_f + hh - w'lambda,  var = h w - w' Lambda w + Rnew.
// It is the same posterior as the reference's reverse-time model in exact arithmetic; the reference's 1e-10 jitter on the predicted
// covariance (lgssm.jl:235) has no counterpart here, so the two agree to ~1e-10 relative, not to the last bit (tests: 1e-8).
struct FusedSmoothArgs {
    int64_t T = 0;
    int64_t step0 = 0, step1 = 0;      // time steps [step0, step1) of this launch, walked from step1 - 1 down to step0
    int d = 0, p = 0, Pq = 0;
    const double *A = nullptr, *H = nullptr, *h = nullptr;
    int64_t sA = 0, sH = 0, sh = 0;
    const double* m_f = nullptr;       // [T][d] filtering means
    const double* P_f = nullptr;       // [T][d*d] filtering covariances (column-major)
    const double* aux = nullptr;       // [T][p][d + 2] (dk_fused_filter)
    const double* Rnew = nullptr;      // [T][p] or [p]
    int64_t sRn = 0;
    double* adj = nullptr;             // Lambda (DP x DP, row-major with ld DP) then lambda (DP): carried between launches
    int first = 1;                     // the launch that starts at the end of the series: (lambda, Lambda) = 0
    double* mean_out = nullptr;        // [T][p]
    double* var_out = nullptr;         // [T][p]
};

template <int DP>
struct FusedSmoothCfg {
    static constexpr int NT = DP / 16, KS = DP / 4, LD = DP + 4, NG = 256 / DP;
    // LDS (doubles): Lambda | T = Lambda A | P_f | lambda | lambda' | m_f | w | z | K | q | partial sums | H rows | scalars
    static constexpr int oL = 0, oT = oL + DP * LD, oPf = oT + DP * LD, ol = oPf + DP * LD, ol2 = ol + DP, oM = ol2 + DP, oW = oM + DP, oZ = oW + DP,
                         oK = oZ + DP, oQ = oK + DP, oRed = oQ + DP, oH = oRed + NG * DP, oS = oH + 16 * DP, oAux = oS + 64,
                         TOTAL = oAux + 16 * (DP + 2);
    static constexpr int APT = (16 * (DP + 2) + 255) / 256;      // aux values per thread (prefetch registers)
    static constexpr size_t LDS_BYTES = (size_t)TOTAL * sizeof(double);
};

template <int DP>
__global__ __launch_bounds__(256) void dk_fused_smooth(const FusedSmoothArgs g) {
    using C = FusedSmoothCfg<DP>;
    constexpr int NT = C::NT, KS = C::KS, LD = C::LD, NG = C::NG;
    extern __shared__ double lds[];
    double* sL = lds + C::oL;
    double* sT = lds + C::oT;
    double* sPf = lds + C::oPf;
    double* sl = lds + C::ol;
    double* sl2 = lds + C::ol2;
    double* smf = lds + C::oM;
    double* sw = lds + C::oW;
    double* sz = lds + C::oZ;
    double* sK = lds + C::oK;
    double* sq = lds + C::oQ;
    double* red = lds + C::oRed;
    double* sH = lds + C::oH;
    double* ss = lds + C::oS;      // [0..3] scalars of the current update; [16..31] h; [32..47] Rnew of the step
    double* sAux = lds + C::oAux;  // the step's p records (v (d), s, nu) of the forward pass
    constexpr int APT = C::APT;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const int d = g.d;
    const int naux = g.p * (d + 2);
    const bool H_shared = g.sH == 0 && g.sh == 0;

    for (int e = tid; e < DP * LD; e += 256) sL[e] = 0.0, sPf[e] = 0.0;
    if (tid < DP) sl[tid] = 0.0, smf[tid] = 0.0, sw[tid] = 0.0, sK[tid] = 0.0;
    __syncthreads();
    if (!g.first) {
        for (int e = tid; e < DP * DP; e += 256) sL[(e / DP) * LD + e % DP] = g.adj[e];
        if (tid < DP) sl[tid] = g.adj[DP * DP + tid];
    }
    double bf[NT][KS];            // bf[J][ks] = A[4 ks + lq][16 J + lr]: B operand of (Lambda A) and A operand of A' (.)
    auto load_A = [&](const double* A) __attribute__((always_inline)) {
#pragma unroll
        for (int J = 0; J < NT; ++J)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) bf[J][ks] = A[(ks * 4 + lq) + (int64_t)(J * 16 + lr) * DP];
    };
    auto load_H = [&](int64_t t) __attribute__((always_inline)) {
        const double* H = g.H + t * g.sH;
        for (int e = tid; e < g.p * DP; e += 256) sH[e] = H[(e / DP) + (int64_t)(e % DP) * g.Pq];
        if (tid < g.p) ss[16 + tid] = g.h[t * g.sh + tid];
    };
    if (g.sA == 0) load_A(g.A);
    if (H_shared) load_H(0);
    // the filtering state of the next (earlier) step travels one step ahead in registers
    constexpr int PPT = (DP * DP) / 256;      // covariance elements per thread
    double pf_n[PPT];
    double mf_n = 0.0, rn_n = 0.0;
    double ax_n[APT];
    auto fetch = [&](int64_t t) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < APT; ++u) {
            const int e = tid + u * 256;
            ax_n[u] = e < naux ? g.aux[t * (int64_t)naux + e] : 0.0;
        }
        if (tid < g.p) rn_n = g.Rnew[g.sRn ? t * g.p + tid : tid];
        const double* P = g.P_f + t * (int64_t)d * d;
#pragma unroll
        for (int u = 0; u < PPT; ++u) {
            const int e = tid + u * 256, i = e % DP, j = e / DP;
            pf_n[u] = (i < d && j < d) ? P[i + (int64_t)j * d] : 0.0;
        }
        if (tid < DP) mf_n = tid < d ? g.m_f[t * d + tid] : 0.0;
    };
    fetch(g.step1 - 1);
    __syncthreads();

    // matvec with a symmetric LDS matrix: out_i = sum_k M[k][i] x[k], K-sliced over the threads, summed by the first DP threads
    auto matvec = [&](const double* M, const double* x, double* out) __attribute__((always_inline)) {
        const int i = tid % DP, slc = tid / DP;
        if (slc < NG) {
            double s = 0.0;
            for (int k = slc; k < DP; k += NG) s += M[k * LD + i] * x[k];
            red[slc * DP + i] = s;
        }
        lds_barrier();
        if (tid < DP) {
            double v = 0.0;
#pragma unroll
            for (int q = 0; q < NG; ++q) v += red[q * DP + tid];
            out[tid] = v;
        }
        lds_barrier();
    };

    for (int64_t t = g.step1 - 1; t >= g.step0; --t) {
        // ---- filtering state of step t -> LDS; prefetch step t - 1
#pragma unroll
        for (int u = 0; u < PPT; ++u) {
            const int e = tid + u * 256;
            sPf[(e % DP) * LD + e / DP] = pf_n[u];
        }
        if (tid < DP) smf[tid] = mf_n;
#pragma unroll
        for (int u = 0; u < APT; ++u) {
            const int e = tid + u * 256;
            if (e < naux) sAux[e] = ax_n[u];
        }
        if (tid < g.p) ss[32 + tid] = rn_n;
        if (t > g.step0) fetch(t - 1);
        if (!H_shared) load_H(t);
        lds_barrier();
        // ---- emission marginals of step t under the smoothed state
        for (int j = 0; j < g.p; ++j) {
            matvec(sPf, sH + j * DP, sw);          // w = P_f h_j'
            matvec(sL, sw, sz);                     // z = Lambda w
            if (w == 0) {
                double hi = 0.0, wi = 0.0, zi = 0.0, li = 0.0, mi = 0.0;
                if (lane < DP) {
                    hi = sH[j * DP + lane];
                    wi = sw[lane];
                    zi = sz[lane];
                    li = sl[lane];
                    mi = smf[lane];
                }
                const double hw = wave_sum(hi * wi), wz = wave_sum(wi * zi), wl = wave_sum(wi * li), hm = wave_sum(hi * mi);
                if (lane == 0) {
                    g.mean_out[t * g.p + j] = hm + ss[16 + j] - wl;
                    g.var_out[t * g.p + j] = hw - wz + ss[32 + j];
                }
            }
        }
        // ---- the p scalar updates of step t, backwards
        for (int j = g.p - 1; j >= 0; --j) {
            const double* ax = sAux + j * (d + 2);
            if (tid < DP) sK[tid] = tid < d ? ax[tid] : 0.0;                 // v
            if (tid == 0) {
                ss[1] = ax[d];          // s
                ss[2] = ax[d + 1];      // nu
            }
            lds_barrier();
            matvec(sL, sK, sq);                     // q = Lambda v   (K = v / s: the 1 / s factors are applied below)
            if (w == 0) {
                double vi = 0.0, qi = 0.0, li = 0.0;
                if (lane < DP) {
                    vi = sK[lane];
                    qi = sq[lane];
                    li = sl[lane];
                }
                const double sinv = 1.0 / ss[1];
                const double kap = wave_sum(vi * qi) * sinv * sinv;          // K' Lambda K
                const double beta = wave_sum(vi * li) * sinv;                // K' lambda
                if (lane == 0) {
                    ss[0] = sinv;
                    ss[3] = kap + sinv;
                }
                if (lane < DP) sl[lane] = li - sH[j * DP + lane] * (beta + ss[2] * sinv);
            }
            lds_barrier();
            {
                const double sinv = ss[0], c = ss[3];
                const double* hj = sH + j * DP;
                for (int e = tid; e < DP * DP; e += 256) {
                    const int a = e / DP, b = e % DP;
                    sL[a * LD + b] += -(hj[a] * sq[b] + sq[a] * hj[b]) * sinv + c * hj[a] * hj[b];
                }
            }
            lds_barrier();
        }
        if (t == 0) break;       // nothing before the first step needs the adjoints
        // ---- transition t - 1 -> t: Lambda <- A' Lambda A, lambda <- A' lambda
        if (g.sA != 0) load_A(g.A + t * g.sA);
#pragma unroll
        for (int tl = 0; tl < NT * NT; ++tl)         // T = Lambda A (Lambda symmetric: its A operand is read by columns)
            if ((tl & 3) == w) {
                const int I = tl / NT, J = tl % NT;
                d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) acc = mfma_f64(sL[(ks * 4 + lq) * LD + I * 16 + lr], bf[J][ks], acc);
#pragma unroll
                for (int r = 0; r < 4; ++r) sT[(I * 16 + lq + 4 * r) * LD + J * 16 + lr] = acc[r];
            }
        lds_barrier();
#pragma unroll
        for (int tl = 0; tl < NT * (NT + 1); ++tl)   // [Lambda | lambda] <- A' [T | lambda]
            if ((tl & 3) == w) {
                const int I = tl / (NT + 1), J = tl % (NT + 1);
                d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    double b;
                    if (J < NT) b = sT[(ks * 4 + lq) * LD + J * 16 + lr];
                    else b = lr == 0 ? sl[ks * 4 + lq] : 0.0;
                    acc = mfma_f64(bf[I][ks], b, acc);
                }
                if (J < NT) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) sL[(I * 16 + lq + 4 * r) * LD + J * 16 + lr] = acc[r];
                } else if (lr == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) sl2[I * 16 + lq + 4 * r] = acc[r];
                }
            }
        lds_barrier();
        if (tid < DP) sl[tid] = sl2[tid];
        lds_barrier();
    }
    __syncthreads();
    if (g.adj && g.step0 > 0) {
        for (int e = tid; e < DP * DP; e += 256) g.adj[e] = sL[(e / DP) * LD + e % DP];
        if (tid < DP) g.adj[DP * DP + tid] = sl[tid];
    }
}
