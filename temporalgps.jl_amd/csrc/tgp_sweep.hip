// The sweep engine: see tgp_sweep.hpp (what and why) and tgp_sweep_body.hpp (the per-lane arithmetic, shared with tests/hostsim).
// gfx950 only.  One wave per workgroup, one wave per SIMD (the kernel is bound by its fp64 instruction stream, not by memory: every
// lane carries the full covariance recursion of its chunk), 4 workgroups per CU by their LDS (the filtering states of a block).
#include "tgp_sweep.hpp"
#include "tgp_alloc.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "tgp_sweep_body.hpp"

namespace tgp_sweep {

__device__ __forceinline__ double shfl_up1(double v) { return __shfl_up(v, 1, 64); }
__device__ __forceinline__ double shfl_dn1(double v) { return __shfl_down(v, 1, 64); }

template <int D> __device__ __forceinline__ bool state_finite(const State<D>& x) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) s += ::fabs(x.m[k]);
#pragma unroll
    for (int k = 0; k < SD<D>::DS; ++k) s += ::fabs(x.P[k]);
    return s < 1e300;      // (false for NaN as well)
}

template <int D, bool SDE, int XS, bool POST>
__global__ __launch_bounds__(64, 1) void k_sweep(const KArgs<D> by_value) {
    (void)by_value;
    const KArgs<D>& ka = *(const KArgs<D>*)__builtin_amdgcn_kernarg_segment_ptr();      // (read in place, scalar loads: tgp_modal.hip k_steady_one)
    constexpr int B = Geo<D>::B, NS = SD<D>::NS, DS = SD<D>::DS;
    constexpr int kStates = B * NS * 64, kTiles = 2 * 64 * (B + 1);      // the block's filtering states; later in a block, its two output tiles
    __shared__ double sF[POST ? (kStates > kTiles ? kStates : kTiles) : 64];
    const int lane = threadIdx.x;
    const long long wave = blockIdx.x;
    const long long T = ka.T;
    const int C = ka.C;
    const long long c = wave * kOwned + lane - 1;
    const bool active = c >= 0 && c < ka.nchunks;
    const long long t0 = active ? c * C : 0;
    long long t1 = active ? t0 + C : 0;
    t1 = t1 < T ? t1 : (active ? T : 0);
    const long long t1r = (t1 + 7) & ~7ll;                    // the runs work in whole blocks: the steps behind the series' end are missing ones
    const bool runs = active && lane >= 1;                    // lane 0 only warms up for lane 1
    const bool owned = runs && lane <= kOwned;                // lane 63 only warms up (backwards) for lane 62
    bool ok = true;

    ModelR<D, SDE> mr;
    mr.init(ka.mc);
    State<D> gen, x0;
    set_state<D>(gen, ka.mc.gm, ka.mc.gP);
    set_state<D>(x0, ka.mc.x0m, ka.mc.x0P);
    double* ck = POST ? ka.ckpt + (size_t)wave * (size_t)(C / B) * NS * 64 : nullptr;

    // ---- forwards.  Pass 0: the last W steps of the chunk from the stationary prior (from x0 where they reach the series' first step); its
    // end state is the NEXT lane's start state.  Pass 1: the chunk from the previous lane's end state (checkpoints, log marginal likelihood).
    State<D> x, e1;
    LmlAcc acc;
    // (POST) the smoothing state of the chunk's first step as its first Wb steps give it -- what the PREVIOUS lane continues from -- is composed
    // during the forward run (RevAcc): from the filtering state behind those steps (the smoothing state where they reach the series' last step)
    const long long te = (t0 + ka.Wb < t1) ? t0 + ka.Wb : t1;
    RevAcc<D> rev;
    rev.reset();
    State<D> b1 = gen;
    {
        const long long tw = t1r - ka.W;
        x = tw <= 0 ? x0 : gen;
        // segment 0: the warm-up; 1: (POST) the chunk's first Wb steps, with the reverse-time composition; 2: the rest of the chunk
        const int nwin = POST ? ka.Wb / B : 0;
        for (int seg = 0; seg < 3; ++seg) {
            if (seg == 1) {
                if constexpr (POST) forward_run<D, SDE, XS, B, true>(ka, mr, t0, nwin, t0, runs ? t1r : t0, x, acc, true, ck, lane, ok, &rev, t0, te, &b1);
            } else {      // (ONE call site for both plain runs: the loop body exists once)
                const bool warm = seg == 0;
                const long long ts = warm ? tw : t0 + (long long)nwin * B;
                forward_run<D, SDE, XS, B>(ka, mr, ts, warm ? ka.W / B : C / B - nwin, warm ? (tw > 0 ? tw : 0) : t0, warm ? t1r : (runs ? t1r : t0), x, acc, !warm,
                                           (POST && !warm) ? ck + (size_t)nwin * NS * 64 : (double*)nullptr, lane, ok);
                if (warm) {
                    e1 = x;
#pragma unroll
                    for (int k = 0; k < D; ++k) x.m[k] = shfl_up1(e1.m[k]);
#pragma unroll
                    for (int k = 0; k < DS; ++k) x.P[k] = shfl_up1(e1.P[k]);
                    if (t0 == 0) x = x0;
                    acc = LmlAcc();
                }
            }
        }
    }
    double dist_f = 0.0, dist_b = 0.0;
    bool finite = true;
    if (owned) {
        dist_f = state_distance<D>(ka.mc, x, e1);
        finite = state_finite<D>(x) && state_finite<D>(e1);
    }

    if (POST) {
        // ---- backwards: the chunk from the next lane's smoothing state of ITS first step; mean and variance of every step.
        State<D> xs;
#pragma unroll
        for (int k = 0; k < D; ++k) xs.m[k] = shfl_dn1(b1.m[k]);
#pragma unroll
        for (int k = 0; k < DS; ++k) xs.P[k] = shfl_dn1(b1.P[k]);
        backward_run<D, SDE, XS, B>(ka, mr, t0, C / B, owned ? t1 : t0, t1 == T, xs, true, ck, sF, lane, ok);
        if (owned) {
            dist_b = state_distance<D>(ka.mc, xs, b1);
            finite = finite && state_finite<D>(xs) && state_finite<D>(b1);
        }
    }

    // ---- the wave's share: fixed-order sums over the lanes
    double lml = owned ? acc.total() : 0.0;
    finite = finite && (::fabs(lml) < 1e300);
    unsigned bits = 0;
    if (owned && !(dist_f <= ka.mc.tol)) bits |= 1u;
    if (owned && POST && !(dist_b <= ka.mc.tol_b)) bits |= 2u;
    if (!ok) bits |= 4u;
    if (!finite) bits |= 8u;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        lml += __shfl_xor(lml, off, 64);
        bits |= (unsigned)__shfl_xor((int)bits, off, 64);
        const double of = __shfl_xor(dist_f, off, 64), ob = __shfl_xor(dist_b, off, 64);
        dist_f = (of > dist_f) ? of : dist_f;
        dist_b = (ob > dist_b) ? ob : dist_b;
    }
    if (lane == 0) {
        double* p = ka.part + (size_t)wave * 4;
        p[0] = lml;
        p[1] = (double)bits;
        p[2] = dist_f;
        p[3] = dist_b;
    }
}

// ------------------------------------------------------------------------------------------------------------------------ host side
struct Engine {
    Plan p;
    Forced forced;
    double* part = nullptr;             // pinned: the waves' shares
    size_t part_cap = 0;
    void* ckpt = nullptr;
    size_t ckpt_cap = 0;
};

Engine* create() { return new Engine(); }
void destroy(Engine* e) {
    if (!e) return;
    if (e->part) (void)tgp_alloc::host_free(e->part);
    if (e->ckpt) (void)tgp_alloc::dev_free(e->ckpt);
    delete e;
}
void force_geometry(Engine* e, int C, int W, int Wb) {
    e->forced.C = C;
    e->forced.W = W;
    e->forced.Wb = Wb;
}
void geometry(const Engine* e, int* C, int* W, int* Wb, int64_t* nwaves) {
    if (C) *C = e->p.C;
    if (W) *W = e->p.W;
    if (Wb) *Wb = e->p.Wb;
    if (nwaves) *nwaves = e->p.nwaves;
}
bool plan(Engine* e, const ModelHost& m, int64_t T, int w_hint, int wb_hint, int num_cu, std::string* why) {
    return make_plan(&e->p, e->forced, m, T, w_hint, wb_hint, num_cu, why);
}

namespace {

template <int D, bool SDE, int XS, bool POST> void launch(Engine* e, hipStream_t stream, const Call& c) {
    KArgs<D> ka;
    std::memcpy(&ka.mc, e->p.mc, sizeof ka.mc);
    ka.st.y = c.y;
    ka.st.mask = c.mask;
    ka.st.R = c.R;
    ka.st.hh = c.hh;
    ka.st.tau = c.tau;
    ka.st.Rnew = c.Rnew;
    ka.st.rnew_per_step = c.rnew_per_step;
    ka.T = c.T;
    ka.C = e->p.C;
    ka.W = e->p.W;
    ka.Wb = e->p.Wb;
    ka.nchunks = e->p.nchunks;
    ka.mean = c.mean;
    ka.var = c.var;
    ka.ckpt = static_cast<double*>(e->ckpt);
    ka.part = e->part;
    hipLaunchKernelGGL((k_sweep<D, SDE, XS, POST>), dim3((unsigned)e->p.nwaves), dim3(64), 0, stream, ka);
}

template <int D, bool SDE, int XS> void launch_x(Engine* e, hipStream_t stream, const Call& c) {
    if (c.mean != nullptr) launch<D, SDE, XS, true>(e, stream, c);
    else launch<D, SDE, XS, false>(e, stream, c);
}
template <int D, bool SDE> void launch_s(Engine* e, hipStream_t stream, const Call& c) {
    const int xs = (c.R != nullptr ? 1 : 0) | (c.hh != nullptr ? 2 : 0);
    switch (xs) {
        case 0: launch_x<D, SDE, 0>(e, stream, c); break;
        case 1: launch_x<D, SDE, 1>(e, stream, c); break;
        case 2: launch_x<D, SDE, 2>(e, stream, c); break;
        default: launch_x<D, SDE, 3>(e, stream, c); break;
    }
}
template <int D> void launch_d(Engine* e, hipStream_t stream, const Call& c) {
    if (e->p.sde) launch_s<D, true>(e, stream, c);
    else launch_s<D, false>(e, stream, c);
}

}  // namespace

const char* kernel_name(int, bool sde, bool post) {
    return sde ? (post ? "k_sweep<sde,posterior>" : "k_sweep<sde,logpdf>") : (post ? "k_sweep<lti,posterior>" : "k_sweep<lti,logpdf>");
}

int enqueue(Engine* e, hipStream_t stream, const Call& c, const char** kname, std::string* err) {
    auto fail = [&](const char* what, hipError_t rc) {
        if (err) *err = std::string(what) + ": " + hipGetErrorString(rc);
        return 1;
    };
    const Plan& p = e->p;
    const bool post = c.mean != nullptr;
    const size_t need_part = (size_t)p.nwaves * 4 * sizeof(double);
    if (need_part > e->part_cap) {
        if (e->part) (void)tgp_alloc::host_free(e->part);
        e->part = nullptr;
        e->part_cap = 0;
        const size_t cap = std::max<size_t>(need_part, 1 << 16);
        hipError_t rc = tgp_alloc::host_malloc((void**)&e->part, cap, hipHostMallocDefault);
        if (rc != hipSuccess) return fail("hipHostMalloc", rc);
        e->part_cap = cap;
    }
    if (post) {
        const int B = p.d <= 3 ? 8 : 4, NS = p.d + p.d * (p.d + 1) / 2;
        const size_t need = (size_t)p.nwaves * (size_t)(p.C / B) * NS * 64 * sizeof(double);
        if (need > e->ckpt_cap) {
            if (e->ckpt) (void)tgp_alloc::dev_free(e->ckpt);
            e->ckpt = nullptr;
            e->ckpt_cap = 0;
            hipError_t rc = tgp_alloc::dev_malloc(&e->ckpt, need);
            if (rc != hipSuccess) return fail("hipMalloc", rc);
            e->ckpt_cap = need;
        }
    }
    if (kname) *kname = kernel_name(p.d, p.sde, post);
    switch (p.d) {
        case 1: launch_d<1>(e, stream, c); break;
        case 2: launch_d<2>(e, stream, c); break;
        case 3: launch_d<3>(e, stream, c); break;
        default: launch_d<4>(e, stream, c); break;
    }
    hipError_t rc = hipGetLastError();
    if (rc != hipSuccess) return fail("k_sweep launch", rc);
    return 0;
}

double finish(Engine* e, int* status, int* w, int* wb, double* dist_f, double* dist_b) {
    double lml = 0.0, df = 0.0, db = 0.0;
    unsigned bits = 0;
    for (int64_t i = 0; i < e->p.nwaves; ++i) {      // fixed order: the same sum for the same geometry
        const double* q = e->part + (size_t)i * 4;
        lml += q[0];
        bits |= (unsigned)q[1];
        df = std::max(df, q[2]);
        db = std::max(db, q[3]);
    }
    if (status) *status = (int)bits;
    if (w) *w = e->p.W;
    if (wb) *wb = e->p.Wb;
    if (dist_f) *dist_f = df;
    if (dist_b) *dist_b = db;
    return lml;
}

}  // namespace tgp_sweep
