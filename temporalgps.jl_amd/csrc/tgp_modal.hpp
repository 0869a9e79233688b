// Stationary-gain engine, ONE-LAUNCH path (round 4): logpdf and posterior marginals of an LTI model with one noise variance,
// scalar observations and no missing data -- the reference's `Fill` layout for RegularSpacing inputs (src/gp/lti_sde.jl:148-160) -- as
//     host plan (tgp_steady_plan.hpp: covariance recursion to its fixed point, gains, variance tables, modal form; a few microseconds)
//   + ONE kernel over y (k_steady_one: reads y once, writes mean and var once; no pass before it, no carry kernel, no reduction kernel)
//   + the host's sum of the workgroups' partial sums of squares (they land in pinned host memory; the call synchronises anyway).
// It computes what tgp_steady.hip computes (same re-association of lgssm.jl:99-238, lgc.jl:46-52,247-257), in the modal coordinates of
// the stationary closed loop.  Applies when the plan says so (tgp_plan::Info::why == kOk); every other case stays on tgp_steady.hip /
// the general engine.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "tgp_steady_plan.hpp"

namespace tgp_modal {

struct Call {
    long long T = 0;
    const double* y = nullptr;      // device
    const double* Rnew = nullptr;   // device: one value, or T values when rnew_per_step
    int rnew_per_step = 0;
    double *mean = nullptr, *var = nullptr;      // device; nullptr: logpdf only
    // A time segment of the series (one rank of several): y, Rnew (per step), mean, var hold the steps [seg_lo, seg_hi) only; yl / yr the
    // `halo` observations in front of / behind them (device; unused at the series' ends).  Default: the whole series.
    long long seg_lo = 0, seg_hi = -1;
    const double *yl = nullptr, *yr = nullptr;
};

struct Engine;
Engine* create();
void destroy(Engine*);
// Host half: builds the plan of a T-step call of the model (nothing is kept from earlier calls).  false: the path does not apply (last_plan().why).
// logpdf_only: the call will be a logpdf over the whole series (no outputs, no segment) -- served by the streaming kernel of tgp_lml.hip.
bool plan(Engine*, const tgp_plan::ModelHost&, long long T, bool logpdf_only = false);
// TGP_OPT_STREAM_MIN_T: -1 the measured crossovers, 0 always, else the series length from which the streaming kernels serve
void set_stream_min_T(Engine*, long long min_T);
// the host's wait for a logpdf-only call's kernel: true once its last workgroup has said so through pinned memory (false: use the stream)
bool await_done(Engine*);
// Enqueues the kernel of the planned call on `stream` (no synchronisation); *kname names it for the profile.
int enqueue(Engine*, hipStream_t stream, const Call&, const char** kname, std::string* err);
// Right behind enqueue: the tables half of the plan when plan() left it for now (the kernel's head wave and last tiles wait for it).
// false: that half declined -- synchronise, discard the outputs, run the call elsewhere.
// host plans of the filter / posterior / rand one-launch paths (defined in tgp_modal.hip only: one set of host compile flags, one CPU check)
void plan_filter(const tgp_plan::ModelHost& m, long long T, tgp_plan::FilterPlan& fp);
void plan_filter_head(const tgp_plan::ModelHost& m, const tgp_plan::FilterPlan& fp, const double* y, double* mout, double* Pout, double* mu_end, double* quad);
bool plan_posterior_head(const tgp_plan::ModelHost& m, const tgp_plan::FilterPlan& fp, const double* y, double* Gh, double* gh, double* Lh, double* Gss,
                         double* Lss, double* mu_end, double* quad);
void plan_rand(const tgp_plan::ModelHost& m, tgp_plan::RandPlan& rp);
bool complete(Engine*, long long T);
// Leaving between enqueue and complete (an error path): raise the flags the kernel may be waiting for, with the failure bit.  Harmless otherwise.
void abandon(Engine*);
// Once the stream has passed the kernel: the log marginal likelihood (a call over the whole series) ...
double finish(const Engine*, long long T);
// ... or the launch's share of the quadratic form (a segment): lml = -(T log 2pi + LS + (T - n0) logS + sum head_quad + iS sum ssq) / 2
void finish_parts(const Engine*, double* ssq, double* head_quad);
// the whole plan (both halves) without an engine: the pure host function tgp_steady_plan of the ABI
tgp_plan::Info plan_only(const tgp_plan::ModelHost&, long long T, tgp_plan::Modal&, tgp_plan::HeadTables&);
const tgp_plan::Info& last_plan(const Engine*);
const tgp_plan::Modal& last_modal(const Engine*);
// the kernel variant plan() chose for the call (profile label)
const char* kernel_name(const Engine*, bool posterior);
const char* kernel_name(const Engine*, const Call&);      // ... for this very call (pointer alignment and segment bounds are part of the choice)
void choose_geometry(int d, int halo, int* waves, int* steps_per_lane);

// rand of an LTI model with the draws supplied (lgssm.jl:65-91; Forward, scalar observations, d <= tgp_plan::kRandMaxD): ONE kernel over the
// draws -- eps_t [T][d], eps_e [T] read once, y [T] written once -- by the same span / halo / in-tile-scan structure, on the dense powers
// of the open-loop transition (tgp_plan::build_rand).  x0: the drawn initial state (host).  Enqueues on `stream`; 0 or a hipError_t.
// _filter of an LTI model behind its head (tgp_plan::build_filter / filter_head; d <= tgp_plan::kRandMaxD): the steps [nhs, T) in ONE kernel --
// y read once, the filtered means [T][d] and covariances [T][d d] written once (either may be nullptr), sum r^2 per workgroup into `part`
// (pinned host memory, at least filter_workgroups() values).  mu_start: the predicted mean of step nhs (host).
// posterior(model, y) (lgssm.jl:193-221, :231-238) rides on it: behind the head the reverse-time transition G and its noise L are constants
// (Gss, Lss: d d doubles each, column-major, host) -- two more fills -- and g_(t+1) = m_t - G mu_(t+1) is at hand where m_t is; `fin` (pinned
// host memory, d doubles) receives the last filtered mean.  G, L [T][d d], g [T][d] (device); the steps [0, nhs) of G, L and [0, nhs] of g
// are the caller's (tgp_plan::posterior_head).
struct PosteriorOut {
    const double *Gss, *Lss;
    double *G, *g, *L, *fin;
};
long long filter_workgroups(const tgp_plan::FilterPlan& plan, long long T);
int filter_lti(hipStream_t stream, const tgp_plan::FilterPlan& plan, const double* mu_start, const double* y, long long T, double* m_out, double* P_out,
               double* part, const PosteriorOut* posterior = nullptr);
// The device half of d logpdf / d (model blocks) of an LTI model by ONE reverse-time pass behind the head, in ONE kernel (d <= 6):
// forwards mu' = Phi mu + a + (A K) u, backwards psi = Phi' psi + h r / S, and the sums SA = sum psi_{t+1} mu_t', Sa, Sk = sum psi_{t+1} r_t,
// Srm = sum r_t mu_t, Sr, sum r^2 over the steps [nhs, T) -- d^2 + 3 d + 2 values per workgroup into `part` (pinned host memory,
// adjoint_workgroups() x adjoint_sums(d) values), psi at step nhs into psi_out (pinned, d values).  tgp_adjoint::finish does the head and the rest.
constexpr int kAdjointMaxD = 6;
inline int adjoint_sums(int d) { return d * d + 3 * d + 2; }
long long adjoint_workgroups(const tgp_plan::FilterPlan& plan, long long T);
// The head beside the kernel (as SmoothCall's): workgroup 0 copies the head's nhs observations to head_in (pinned) and raises head_in_flag; the
// host runs the head forwards and answers with the predicted mean of step nhs in mu0 (pinned) + mu0_flag; flags hold 2 seq.  mu_start is then unused.
struct HeadHandover {
    double* head_in = nullptr;
    long long* head_in_flag = nullptr;
    const double* mu0 = nullptr;
    const long long* mu0_flag = nullptr;
    long long seq = 0;
};
int adjoint_lti(hipStream_t stream, const tgp_plan::FilterPlan& plan, const double* mu_start, const double* y, long long T, double* part, double* psi_out,
                const HeadHandover* handover = nullptr);
// logpdf + posterior marginals of an LTI model behind its head in ONE kernel on DENSE powers of the closed loop (forwards) and of the settled
// reverse-time transition (backwards): the models the modal plan declines (d <= tgp_plan::kRandMaxD).  The head runs on the host
// (plan_smooth_head_*: forwards before the launch, its data-free tables beside the kernel, backwards behind it from xi_out).
// mean == nullptr: logpdf only.  part: smooth_workgroups() values, xi_out: d values, tvb: tgp_plan::kTailMax values -- pinned host memory.
struct SmoothCall {
    long long T = 0;
    const double *y = nullptr, *Rnew = nullptr;      // device
    int rnew_per_step = 0;
    const double* tvb = nullptr;
    double *mean = nullptr, *var = nullptr;          // device
    double *part = nullptr, *xi_out = nullptr;
    // The head beside the kernel (overlap_allowed()): the launch does not wait for the head's forward recursion, and nothing between the launch
    // and the end of the kernel goes through a stream (a copy on a second stream may share the kernel's hardware queue and wait for it: measured --
    // the kernel would wait for the host, the host for the copy, the copy for the kernel).  Everything is handed over through pinned memory:
    //   head_in / head_in_flag   workgroup 0 copies the head's nhs observations (and, with rnew_per_step, its nhs new noise variances) there
    //                            first thing (nullptr: the caller's inputs are host arrays already);
    //   mu0 / mu0_flag           the host's answer: the predicted mean of step nhs (workgroup 0 waits for it, bounded);
    //   xi_out / xi_flag         raised by workgroup 0 once xi at step nhs is known: the host runs the head backwards;
    //   head_out / head_out_flag the head's nhs means and nhs variances from the host; a workgroup from the middle of the dispatch waits for them (bounded) and writes
    //                            them to mean / var [0, nhs).
    // Flags hold 2 seq once raised.  mu0_flag == nullptr: mu_start by value, head outputs are the caller's to write.
    double* head_in = nullptr;
    const double* mu0 = nullptr;
    const double* head_out = nullptr;
    long long *head_in_flag = nullptr, *xi_flag = nullptr;
    const long long *mu0_flag = nullptr, *head_out_flag = nullptr;
    long long seq = 0;
    // A DRAW from the posterior instead of its marginals (rand of the reverse-time model, lgssm.jl:65-91 / :193-221; d <= kSmoothRandMaxD):
    // eps_t [T][d], eps_e [T] (device), `mean` receives the draw, `var` stays unused; U (d d, row-major, upper): chol(L_settled + 1e-9 I).U,
    // v0 = G xi_T, s0 = h' xi_T with xi_T = chol(P_final + 1e-12 I).U' eps_0 (tgp_plan::smooth_rand_factors).  With the head beside the kernel,
    // head_in also receives the head's eta [nhs] and eps [nhs][d] behind y | Rnew (offsets 2 nhs, 3 nhs).
    // an emission offset per step (device, [T]; a mean function at the inputs: lti_sde.jl:118-131) instead of the plan's one value; the gains do not
    // see it.  With the head beside the kernel head_in receives the head's offsets at offset 10 nhs (the host's head functions take them as `hh_t`).
    const double* hh_t = nullptr;
    const double *eps_t = nullptr, *eps_e = nullptr;
    const double *U = nullptr, *v0 = nullptr;
    double s0 = 0.0;
};
constexpr int kSmoothRandMaxD = 6;
// false when the process runs with synchronous launches (HIP_LAUNCH_BLOCKING, AMD_SERIALIZE_KERNEL, ...) or TGP_MODAL_OVERLAP=0: a kernel
// that waits for a flag the host raises behind the launch would wait for itself
bool overlap_allowed();
// the host's wait for a flag in pinned memory the kernel on `stream` raises (value >= v): returns false once the stream has drained without it
bool await_host_flag(const long long* flag, long long v, hipStream_t stream);
void plan_smooth(const tgp_plan::ModelHost& m, long long T, tgp_plan::SmoothPlan& sp, double* tvb, bool post = true);      // post = false: logpdf only (tvb unused)
void plan_smooth_head_forward(const tgp_plan::ModelHost& m, const tgp_plan::SmoothPlan& sp, const double* y, double* mu_end, double* quad, const double* hh_t = nullptr);
bool plan_smooth_head_tables(const tgp_plan::ModelHost& m, const tgp_plan::SmoothPlan& sp);
void plan_smooth_head_backward(const tgp_plan::ModelHost& m, const tgp_plan::SmoothPlan& sp, const double* y, const double* lam, double* mean, double* vb);
// steps a workgroup owns (its tiles minus the halo in front and -- with the backward half -- behind), and the workgroups of a T-step call
bool plan_smooth_rand_factors(const tgp_plan::SmoothPlan& sp, const double* eps0, double* U, double* v0, double* s0);      // (behind plan_smooth; false: not positive definite)
void plan_smooth_head_backward_rand(const tgp_plan::ModelHost& m, const tgp_plan::SmoothPlan& sp, const double* y, const double* delta, const double* eps_e,
                                    const double* eps_t, const double* rn, bool rn_per_step, double* out);
long long smooth_span(const tgp_plan::SmoothPlan& sp, bool post = true);
long long smooth_workgroups(const tgp_plan::SmoothPlan& sp, long long T, bool post = true);
int smooth_lti(hipStream_t stream, const tgp_plan::SmoothPlan& sp, const double* mu_start, const SmoothCall& c);
int rand_lti(hipStream_t stream, const tgp_plan::RandPlan& plan, const double* x0, const double* eps_t, const double* eps_e, long long T, double* y,
             const char** kname);

}  // namespace tgp_modal
