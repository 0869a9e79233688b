// Stationary-gain engine for WIDE states (8 < d <= 63; round 6): the log marginal likelihood of a Forward LTI model with scalar observations, one
// noise variance and no missing data (lgssm.jl:147-165 on the reference's `Fill` layout) across the whole chip -- what the one-launch kernels of
// tgp_modal.hip do for d <= 8, without a modal form: products of kernels (lti_sde.jl:377-400: ApproxPeriodicKernel() * Matern32Kernel(), d = 28)
// have defective closed loops, so the recursion runs on the DENSE closed-loop matrix.
//
//   host plan   the covariance half of the filter never sees y: iterate it to its fixed point (n0 steps, their gains K_t and innovation variances S_t
//               kept for the head), Phi = (I - K h') A, the observer row g = A' h, and `halo` = the number of steps after which Phi^halo is below 2^-60
//               (repeated squaring);
//   head        the first nhs steps (time-varying gains) on the host, from the head's observations copied back;
//   kernel      the steps behind the head are cut into chunks, one WAVE each: lane i owns component i of the filtered mean and row i of Phi in
//               registers, the state goes round through a 512-byte LDS line (one ds_write, d/2 broadcast ds_read_b128 per step), lane d -- the
//               observer -- carries the row -g and so computes the innovation r_t by the same multiply-adds.  A chunk starts `halo` steps early
//               from a zero state (the closed loop forgets it to 2^-60) and sums r_t^2 over its own steps only.
//   posterior   marginals(replace_observation_noise_cov(posterior(model, y), Rnew)) (lgssm.jl:99-115, 193-238) in Bryson-Frazier form: the forward
//               kernel keeps its innovations (8 B per step), a second kernel of the same shape runs lam_t = h r_t / S + Psi lam_(t+1) backwards
//               (Psi = (I - h K') A'), its observer row gw = R A K gives mean_t = y_t - (R / S) r_t + gw . lam_(t+1); the variance is a constant
//               between the head (host, from Lam_inf backwards through the head's steps) and the last n1 steps (a data-free table).
// Before this engine such models ran on ONE compute unit (tgp_dense_fused.hpp: a persistent kernel, sequential in time).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

namespace tgp_wide {

constexpr int kMaxD = 63;          // (lane d is the observer)
constexpr int kHeadMax = 8192;     // steps until the covariance recursion must have settled
constexpr int kTailMax = 8192;     // steps at the series' end whose smoothed variance is still in its transient
constexpr int kMaxChunks = 4096;   // (four times as many where the series is long enough: tgp_wide.hip plan)

struct Engine;

struct ModelHost {      // shared blocks, column-major as handed to tgp_model_set
    int d = 0;
    const double *A = nullptr, *a = nullptr, *Q = nullptr, *H = nullptr;
    double hh = 0.0, R = 0.0;
    const double *x0m = nullptr, *x0P = nullptr;
};

enum Why { kOk = 0, kNotPD = 1, kNotSettled = 2, kSlowMixing = 3, kTooShort = 4, kAlloc = 5, kTailLong = 6 };
struct Info {
    int why = kOk, n0 = -1, nhs = 0, halo = 0;
    int why_post = kOk, n1 = -1, halo_back = 0;
    long long chunks = 0, chunk_len = 0;
    double plan_ms = 0.0, plan_post_ms = 0.0;      // 0 when the plan of the previous call was kept
};

struct Call {      // device pointers; mean == nullptr: logpdf only
    long long T = 0;
    const double* y = nullptr;
    const double* Rnew = nullptr;      // [T] or [1]
    int rnew_per_step = 0;
    double *mean = nullptr, *var = nullptr;
    double *fm = nullptr, *fP = nullptr;      // _filter (lgssm.jl:171-187): filtered means [T][d] and covariances [T][d d], both or none (with mean == nullptr)
    const double* h_t = nullptr;       // [T]: an emission offset per step (a mean function at the inputs) in place of ModelHost::hh, which is then 0
};

Engine* create();
void destroy(Engine* e);
inline bool supports(int d) { return d > 8 && d <= kMaxD; }      // (d <= 8: the modal engine's; 9 <= d <= 16 otherwise run the general chunked scan)
// The plan of model `m` for a series of T steps (kept between calls while the model's blocks and T stand).  false: the engine does not apply (Info::why).
bool plan(Engine* e, const ModelHost& m, long long T);
// ... and its posterior half (the backward recursion's matrix, the variances of the series' two ends), built once per planned model.  false: Info::why_post.
bool plan_posterior(Engine* e, long long T);
const Info& last_plan(const Engine* e);
// logpdf (and, with Call::mean, the posterior marginals: lgssm.jl:99-115 on posterior(model, y) with the noise replaced by Rnew) of the planned model:
// the head on the host, ONE kernel behind it (two for the posterior: forward keeping the innovations, backward in Bryson-Frazier form).
// Synchronises `stream`.  0, or a hipError_t.
int run(Engine* e, hipStream_t stream, const Call& c, double* lml_out, std::string* err);
const char* kernel_name(const Engine* e);
bool filter_ready(const Engine* e);
// the planned model's stationary gain K [d], innovation variance S, and (behind plan_posterior) the smoothed emission variance's two parts: vbase - qinf
void stationary(const Engine* e, double* K, double* S, double* vbase, double* qinf);      // the plan kept the head's filtered covariances (64 MB at most)
// rand(model) with the draws supplied (lgssm.jl:65-91): x0_host the drawn initial state (host, d), eps_t [T][d], eps_e [T], y_out [T] device pointers.
// Enqueues ONE kernel (k_wide_rand) on `stream` -- no synchronisation behind it.  *declined: the open loop does not forget (nothing was enqueued).
int rand(Engine* e, hipStream_t stream, const ModelHost& m, long long T, const double* x0_host, const double* eps_t, const double* eps_e, double* y_out, bool* declined,
         std::string* err);

}  // namespace tgp_wide
