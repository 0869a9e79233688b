// Stationary-gain scan engine (round 3): logpdf and posterior marginals of an LTI model with ONE noise variance, scalar
// observations and no missing data -- the reference's `Fill` layout for RegularSpacing inputs (src/gp/lti_sde.jl:148-160), i.e. every
// BASELINE configuration of the scan engine.
//
// For such a model the covariance half of the Kalman recursion (predict lgc.jl:46-52, update lgc.jl:247-257, invert_dynamics
// lgssm.jl:231-238, the Reverse step_marginals lgssm.jl:111-115) never sees the data.  k_setup runs it once per call, sequentially,
// until the filtered covariance stops changing (n0 steps, ~60 for the bench model) and tabulates the per-step gains of that head,
// the stationary gains, and the smoothed variances of the head and of the last n1 steps (where the smoothed covariance is still in its
// transient from the final filtered state).  What is left for the T steps is the MEAN half, two linear recursions
//     forward   r_t = y_t - h - H mu_t,            mu_{t+1} = A mu_t + a + (A K_t) r_t            (mu_t = predicted mean)
//     backward  mean_t = y_t - (R / S_t) r_t + H lam,   lam <- G_t lam + (G_t K_t) r_t            (lam = smoothed - filtered mean)
// with constant coefficients outside the head: a wave owns a tile of 512 consecutive steps (8 per lane, everything in registers),
// scans its lanes with the powers Phi^(8 2^k), G^(8 2^k), and the tiles are chained by one small carry kernel.  Two passes over y
// (tile elements, then outputs), no scratch of size T.  logpdf = -(T log 2pi + sum log S_t + sum r_t^2 / S_t) / 2.
//
// Results agree with the reference recursion to rounding (the operations are re-associated, not approximated: the covariance is
// iterated to the point where it no longer changes by more than 2 ulp); tolerance against the oracle as for every other path:
// logpdf 1e-10 relative, marginals 1e-8.  When the path does not apply (the covariance does not settle within kHeadMaxTiles tiles,
// or the series is shorter than head + tail) every kernel exits early, result[6] reports it, and the caller runs the general path.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

namespace tgp_steady {

constexpr int kMaxD = 8;
constexpr int kTile = 512;         // steps per wave
constexpr int kSub = 8;            // steps per lane
constexpr int kHeadMaxTiles = 4;   // the head (per-step gains) may span this many tiles: n0 < 2048
constexpr int kTailMax = 2048;     // ... and the smoothed covariance's transient at the end this many steps
constexpr int kPowMax = 48;        // powers Phi^(2^k), G^(2^k), k < kPowMax

// result record (device, 8 doubles, shared with the general path): [0] lml, [2] not-PD count, [6] status, [7] n0
constexpr double kStatusRan = 1.0, kStatusNotApplicable = 2.0;

struct Hooks {       // per-launch profiling brackets (tgp_api.hip: LaunchScope)
    void* ctx = nullptr;
    void (*begin)(void*, const char*) = nullptr;
    void (*end)(void*) = nullptr;
};

struct ModelDev {    // shared blocks of the bound model, device pointers, column-major as handed to tgp_model_set
    int d = 0;
    const double *A = nullptr, *a = nullptr, *Q = nullptr, *H = nullptr, *hh = nullptr, *R = nullptr;
    const double* x0 = nullptr;    // packed: m (d), upper triangle of P by columns
};

struct CallDev {
    int64_t T = 0;
    const double* y = nullptr;
    const double* Rnew = nullptr;  // one value, or T values when rnew_per_step
    int rnew_per_step = 0;
    double *mean = nullptr, *var = nullptr;   // nullptr: logpdf only
    double* result = nullptr;
    bool grad = false;             // adjoint call: logpdf + the record behind d logpdf / d (model blocks) (grad_record; mean / var unused)
};

// A time shard of a longer series (tgp_multi.hip): the call runs in two halves around ONE exchange. Half 0 (setup, pass 1, a carry pass
// under a provisional boundary) leaves the segment's element in `slot` (shard_slot_size(d) doubles, device memory); after an all-gather
// of the slots, half 1 folds the chain of elements into this segment's real boundary and runs the carry pass, pass 2 and the reduction.
// `first`: the segment starts the series (it owns the head; the others have stationary steps only and start from the exchanged mean);
// `last`: it ends the series (it owns the smoother's tail; the others hand their end state on and need T % 512 == 0).
struct ShardDev {
    int first = 1, last = 1, post = 0;
    double* slot = nullptr;
    const double* gathered = nullptr;      // [world][shard_slot_size(d)], rank-major
    int world = 1, rank = 0;
};

struct Engine;
Engine* create();
void destroy(Engine*);
bool supports(int d);
// Enqueues the whole call on `stream` (no synchronisation).  Returns 0, or a hipError_t cast to int with *err set.
int enqueue(Engine*, hipStream_t stream, const ModelDev&, const CallDev&, const Hooks&, std::string* err);
int enqueue_shard(Engine*, hipStream_t stream, const ModelDev&, const CallDev&, const ShardDev&, int phase, const Hooks&, std::string* err);
size_t shard_slot_size(int d);
// Adjoint calls (CallDev::grad).  The record (device memory of the engine, grad_record_size(d) doubles, valid once the stream has
// passed the call) holds, in this order: the sums over the stationary tiles  SA [d][d] = sum psi_{t+1} mu_t', Sa [d] = sum psi_{t+1},
// Sk [d] = sum psi_{t+1} r_t, Srm [d] = sum r_t mu_t, Sr = sum r_t, SSQ = sum r_t^2  (psi_{t+1} = d logpdf / d mu_{t+1}, mu_t the predicted
// mean, r_t the innovation);  psi [d] and mu [d] at the end of the head;  n0, head tiles, T, applies (as doubles);  the model blocks
// A [d*d] a [d] Q [d*d] H [d] hh R x0 (packed) as bound.  tgp_api.hip finishes the gradient on the host (head steps + the reverse sweep
// through the n0 steps of the covariance recursion).
size_t grad_record_size(int d);
const double* grad_record(const Engine*);
// diagnostics of the last enqueued call (synchronises the stream): n0, n1, head tiles, applicable
int last_info(Engine*, hipStream_t stream, int64_t out[4]);

}  // namespace tgp_steady
