/* libtgp_hip.so -- C ABI of the MI355X (gfx950) Kalman filter / RTS smoother engine.
 *
 * Drop-in boundary for the linear-Gaussian state-space hot path of TemporalGPs.jl. The reference has no
 * FFI layer; its extension seam is Julia dispatch on the StorageType tag handed to `to_sde`
 * (/root/reference/src/util/storage_types.jl:1, src/gp/lti_sde.jl:12-16) and on the AbstractLGSSM
 * supertype (src/models/lgssm.jl:1). Each entry point below replaces one reference method on an LGSSM;
 * the Julia-side binding a maintainer adds is shown in INTEGRATION.md (julia/TemporalGPsHIP.jl).
 *
 * Conventions
 *   - all reals are fp64, all sizes int64_t, matrices are column-major d x d blocks (Julia layout);
 *   - per-step arrays are [T][...] contiguous; an array flagged TGP_SHARED_* holds ONE block used by all
 *     steps (FillArrays.Fill: what RegularSpacing inputs produce, lti_sde.jl:148-160);
 *   - observations: p == 1 is ScalarOutputLGC (linear_gaussian_conditionals.jl:225-257): H is the d-vector with
 *     emission.A = H', h and R scalars per step. p > 1 is SmallOutputLGC (lgc.jl:113-141) with DIAGONAL noise:
 *     H is [T][p][d] (row j = emission.A[j, :], i.e. Julia's A' column-major), h [T][p], R [T][p] the noise
 *     diagonal, y / missing / mean / var / eps_e are [T][p]. The p observations of a time step are absorbed as
 *     p consecutive scalar updates (algebraically the joint update). A dense R must be whitened by the caller
 *     (H <- L^-1 H, h <- L^-1 h, y <- L^-1 y, R <- I, lml -= sum log diag L); the Python mirror does that.
 *     `marginals` / `posterior_marginals` return the DIAGONAL of the p x p marginal covariance.
 *   - ordering 0 = Forward, 1 = Reverse (gauss_markov_model.jl:1-9,38-40);
 *   - return codes: 0 ok; 1 bad argument / dimension mismatch (lgssm.jl:202-208); 2 not positive
 *     definite (Julia PosDefException / DomainError at lgc.jl:135,250, lgssm.jl:235); 3 HIP runtime error;
 *     4 unsupported. No exception or abort crosses the ABI; tgp_last_error() gives the message.
 *   - ownership: host pointers are read/written during the call only and never retained. Device pointers
 *     (TGP_DEVICE_PTRS / TGP_IN_DEVICE / TGP_OUT_DEVICE) are BORROWED: model arrays must stay alive until
 *     the next tgp_model_set / tgp_destroy, per-call arrays until the call returns.
 *   - threading: calls on one handle are serialised by the caller; every call returns with its results complete
 *     (blocking). Distinct handles may be used from distinct threads. A handle's HIP stream comes from a small
 *     per-device POOL (8 light + 2 heavy streams, handed out round robin: a stream is an HSA queue, and a process
 *     that keeps thousands of models alive must not hold thousands of them): handles k and k + 8 share a stream,
 *     and calls of handles that share one serialise -- on the device, and on the host through the stream's lock,
 *     which a call holds from entry to return (so graph capture, the pinned-memory hand-overs of the one-launch
 *     paths and the caching allocator never see a second thread on their stream). A stream substituted with
 *     tgp_set_stream is outside the pool and its lock: keep it to one thread at a time.
 *   - device inputs (TGP_IN_DEVICE) must be COMPLETE when the call is made: the library's streams do not wait
 *     for the stream that produced them (the Python mirror synchronises torch's current stream before a call on
 *     torch tensors; the Julia glue passes host arrays). Several one-launch paths talk to the host through pinned memory while
 *     their kernel runs (the head of an LTI series is computed on the host beside the kernel); under the
 *     runtime's synchronous-launch switches (HIP_LAUNCH_BLOCKING, AMD_SERIALIZE_KERNEL, ...) they run the same
 *     steps one after the other. Environment switches for A/B runs: TGP_MODAL_OVERLAP=0 (no host / device
 *     hand-over at all), TGP_MODAL_HOSTHEAD=0 (the head of the one-launch kernel inside the kernel).
 */
#ifndef TGP_HIP_H
#define TGP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tgp_handle tgp_handle;

/* ---- flags ------------------------------------------------------------------------------------ */
#define TGP_SHARED_A (1u << 0)
#define TGP_SHARED_a (1u << 1)
#define TGP_SHARED_Q (1u << 2)
#define TGP_SHARED_H (1u << 3)
#define TGP_SHARED_h (1u << 4)
#define TGP_SHARED_R (1u << 5) /* also: Rnew is one scalar in tgp_posterior_marginals */
#define TGP_SHARED_ALL 0x3fu
#define TGP_SMALL_OUTPUT (1u << 8) /* tgp_model_set: emissions are SmallOutputLGC even though p == 1 (rand adds 1e-9 to R) */
#define TGP_DEVICE_PTRS (1u << 16) /* tgp_model_set: A..R are device pointers (borrowed) */
#define TGP_IN_DEVICE (1u << 16)   /* per-call inputs (y, missing, Rnew, eps) are device pointers */
#define TGP_OUT_DEVICE (1u << 17)  /* per-call array outputs are device pointers */
#define TGP_REUSE_REDUCE (1u << 18) /* reuse pass 1 + upward scans of the previous call (same model & y) */

/* ---- return codes ----------------------------------------------------------------------------- */
#define TGP_OK 0
#define TGP_EINVAL 1
#define TGP_ENOTPD 2
#define TGP_EHIP 3
#define TGP_EUNSUPPORTED 4

/* ---- options (tgp_set_option) ------------------------------------------------------------------ */
#define TGP_OPT_CHUNK 1   /* steps per lane in the chunked scan (0 = auto) */
#define TGP_OPT_PROFILE 2 /* 1: bracket every kernel with hipEvents (see tgp_profile_*) */
#define TGP_OPT_VARIANT 3 /* d = 5..8: 0 auto (per operation, the inlined build where it passes the run-time known-answer check
                             against the out-of-line build), 1 out-of-line build only, 2 inlined build for everything,
                             3 out-of-line build + the group-per-chunk logpdf kernels (used by the check itself) */

#define TGP_OPT_GROUP 5 /* d = 5..8 LTI models: logpdf with eight lanes per chunk (tgp_group.hpp) and block scans over filter
                           elements with eight lanes per element (tgp_group_scan.hpp), once they have reproduced the out-of-line
                           build in the run-time check: 1 (default) where they are faster (d >= 7), 2 for every d = 5..8,
                           0 never; + 4 keeps the lane-per-element block scans */
#define TGP_OPT_SPLIT_SMOOTHER 7 /* lane-per-chunk passes: pass 2 of the smoother as two kernels (filter + scratch, then the chunk smoother
                                    elements from the scratch): 1 (default) for d = 6, 7, 2 also for d = 5, 0 never (fused MODE 2 kernel) */
#define TGP_OPT_DENSE_STRUCTURE 8 /* dense path (d > 16): 1 (default) a shared A / H with at most 8 entries per row (what
                                     lgssm_components(::Separable, ...) builds: I (x) A_t, I (x) H_t') is applied in sparse form; 0 the
                                     reference's dense products on the fp64 MFMA GEMM kernels. Set before tgp_model_set. */
#define TGP_OPT_DENSE_FUSED 10 /* dense path, 16 < d <= 64 and p <= 16: 1 (default) the filter pass (logpdf, filtering distributions, prior marginals,
                                  rand) runs as ONE persistent kernel (P in LDS, fp64 MFMA products, scalar updates, tgp_dense_fused.hpp) -- ~3 us per
                                  step instead of ~26 us of dependent launches; posterior marginals keep the reference's RTS chain (with its
                                  1e-10 jitter, lgssm.jl:235). 2: posterior marginals too, by a persistent backward pass in modified
                                  Bryson-Frazier form -- no counterpart of that jitter, variances by a difference: agrees with the chain to
                                  ~1e-6 relative only, hence opt-in. 0: the per-step kernel chain that larger states use, for everything.
                                  Takes effect at the next tgp_model_set. */
#define TGP_OPT_SHARED_PARTS 11 /* scan path, pass 1 (1 default / 0 off / 2): for a Forward model with every block shared (LTI), ONE noise
                                   variance, scalar observations and no missing data, the matrix parts (Abar, C, J) of a chunk's filter
                                   element and its per-step (w, Cv, 1/s) do not depend on the observations: every chunk has the same.
                                   They are tabulated once per bound model and chunk length (one lane, ~150 sequential steps: 1.4 ms
                                   at d = 3, longer than a whole call) and pass 1 then runs only the vector half of the recursion
                                   (d <= 6; bit-identical results). The table is never built on the caller's critical path: the second
                                   eligible call on a bound model launches the build on a side stream and still runs the general pass;
                                   later calls use the table once it is complete. 2 = build it in line on the first call (tests). */
#define TGP_OPT_STEADY 12 /* Forward LTI models (every block shared) with ONE noise variance, scalar observations, no missing data -- the
                             reference's Fill layout for RegularSpacing inputs, lti_sde.jl:148-160. For such a model the covariance half
                             of the Kalman / RTS recursion never sees the data.
                             3 (default since round 4): as 2, with the engine's ONE-LAUNCH form (tgp_modal.hip) tried first: the host runs the
                               covariance recursion to its fixed point inside the call (microseconds in double; see tgp_steady_plan), puts the
                               stationary closed loop into modal form (O(d) per step), and ONE kernel over y does the rest: every workgroup
                               starts `halo` steps early from a zero state and stops `halo` steps late (both mean recursions have forgotten a
                               state by then: |eigenvalue|^halo <= 2^-64), so there is no pass over y before it and no carry between
                               workgroups; the workgroups' partial sums of squares land in pinned host memory and the host adds them.
                               Models it declines (ill-conditioned modal form, slow mixing, long heads) go on as with 2.
                             2: the stationary-gain scan engine (tgp_steady.hip, d <= 8) serves tgp_logpdf and
                               tgp_[logpdf_and_]posterior_marginals: the covariance recursion is run once per call until it no longer
                               changes (n0 steps, ~60 at the bench model; per-step gains of that head are tabulated), and the T steps
                               are left with two linear recursions of the MEAN (forward: innovations; backward: smoothed minus filtered
                               mean) that a wave scans over tiles of 512 steps, two passes over y, no scratch of size T. Re-associated,
                               not approximated: results agree with the sequential recursion to rounding (tolerances as everywhere:
                               logpdf 1e-10 relative, marginals 1e-8). Whether it applies (the covariance settles within 2048 steps, the
                               series is longer than head + tail) is decided on the device inside the call; a call that finds it does
                               not is re-run on the general path and the bound model is remembered as such.
                             1: the general chunked-scan engine, passes 2 and 3 (d <= 3) switching to mean-only steps once a chunk's
                               covariance repeats with period 2 bit for bit (decided per wave at run time by comparing bits, so no
                               result changes in any bit against 0).
                             0: the general engine, every step in full. */
#define TGP_OPT_GRAPH 9 /* hipGraph replay of the launch chain of tgp_logpdf / tgp_[logpdf_and_]posterior_marginals: a call with device
                           pointers that repeats the previous call's arguments is recorded once (stream capture, kernel nodes only)
                           and then replayed with one hipGraphLaunch. 0 (default) off, 1 on, -1 on for T <= 2^20. Measured on
                           BASELINE config 1 (T = 1e4): 84 us per combined call with and without replay -- the chain is bound by
                           the dependent dispatches on the device, not by the host's enqueue -- hence off by default. Any other
                           entry point, option or model change in between drops the recording. */
#define TGP_OPT_SDE_CLOSED_FORM 13 /* models set with tgp_model_set_sde, d <= 8: 1 (default) when the drift matrix is block diagonal with one
                           eigenvalue per block and nilpotency <= 3 (sums of scaled / stretched Matern-1/2, -3/2, -5/2 terms), the passes
                           evaluate A_k = exp(F dt_k) in closed form and Q_k = Pinf - A_k Pinf A_k' in registers from the 8-byte gap dt_k
                           (lti_sde.jl:135-146) instead of reading a tiled [T][2 d^2] record three times; 0: always the tiled record
                           (A/B timing, tests). Other drift matrices and the gradient passes use the tiled record either way. */
#define TGP_OPT_SWEEP 14 /* 1 (default): Forward models with scalar observations and d <= 4 whose GAINS vary in time -- a missing-data mask, a noise
                           variance or an emission offset per step (shared A, a, Q, H), or irregular spacing (tgp_model_set_sde with closed-form
                           transitions) -- run tgp_logpdf / tgp_[logpdf_and_]posterior_marginals on the sweep engine (tgp_sweep.hip):
                           ONE kernel; a lane owns a chunk of consecutive steps and runs the reference's sequential recursion over it in registers;
                           a chunk's start state comes from a warm-up over the steps in front of it (the filter forgets), its smoothing state at
                           the end from the next chunk's backward warm-up; the filtering states a backward step needs are recomputed from
                           checkpoints.  Every hand-over is checked (the run from the handed-over state must reproduce the warm-up's end state
                           to 1e-12 of a state's size): a call whose warm-ups prove too short is repeated with longer ones, a model that mixes
                           too slowly goes to the general engine.  What the reference's predict path produces: posterior_lti_sde.jl:20-37,97-131,
                           missings.jl:25-41, lti_sde.jl:135-146.  0: the general chunked-scan engine as before. */
#define TGP_OPT_SWEEP_CHUNK 15       /* tests: steps per chunk of the sweep engine (0 automatic) */
#define TGP_OPT_SWEEP_WARMUP 16      /* tests: forward warm-up steps (0 automatic); a forced geometry is never repaired by longer warm-ups */
#define TGP_OPT_SWEEP_WARMUP_BACK 17 /* tests: backward warm-up steps (0 automatic) */
#define TGP_OPT_STREAM_MIN_T 18 /* series length from which the STREAMING kernels of the stationary-gain engine serve a call: persistent
                                   waves with ~7 us more fixed latency and 1.3 - 2.7 x the throughput of k_steady_one.  -1 (default): the measured
                                   crossovers (logpdf 5e6, posterior marginals 3e6 steps at d = 3); 0: always (the tests); a length: from there on */
#define TGP_OPT_WIDE 19 /* 1 (default): logpdf of a Forward LTI model with 16 < d <= 63 and scalar observations on the stationary closed loop across the
                           chip (tgp_wide.hip: k_wide_lml); 0: the dense engine's sequential passes (the A/B of the tests) */
#define TGP_OPT_TIMING 6 /* 1: record the hipEvents behind tgp_last_timing (off by default: ~30 us of host time per call) */
#define TGP_OPT_FUSE_SCAN 4 /* 1 (default): the level-0 scan reduce / apply of the forward scan run inside the chunk kernels;
                               0: stand-alone k_scan_reduce / k_scan_apply launches (bit-identical results, for A/B timing) */

/* ---- lifetime ---------------------------------------------------------------------------------- */
int tgp_create(tgp_handle** h, int device);
int tgp_destroy(tgp_handle* h);
const char* tgp_last_error(const tgp_handle* h);
int tgp_set_option(tgp_handle* h, int option, int64_t value);
/* run on a caller-provided hipStream_t (e.g. torch's current stream); NULL restores the handle's own */
int tgp_set_stream(tgp_handle* h, void* hip_stream);
/* the stream the handle's work is enqueued on (its own unless tgp_set_stream replaced it), as a hipStream_t */
int tgp_get_stream(tgp_handle* h, void** hip_stream);
/* hipStreamSynchronize on a stream of the CALLER's: the stream that produced a TGP_IN_DEVICE input must have passed it when the call is
 * made (see "device inputs" above); the Python mirror calls this on torch's current stream in front of every call with device inputs */
int tgp_stream_synchronize(void* hip_stream);
const char* tgp_version(void);
/* Binds the CALLING thread (and the threads it creates from then on) to the CPUs next to `device` (its PCI function's local_cpulist). One process per
 * GPU is the deployment this library is built for; on a two-socket host a thread on the far socket pays a second hop for every host <-> device
 * hand-over of a call (flags in pinned memory, kernel arguments, doorbells): ~13 us of a 0.11 ms headline step. Call it before the first
 * tgp_create so that the handle's pinned memory is first touched on that node. TGP_EUNSUPPORTED: no such information / not allowed -- nothing changed. */
int tgp_bind_host_thread(int device);
/* which build of the kernels the current model runs on: 1 out-of-line (safe), 2 fully inlined (d = 5, 6 after the check);
   dense path (d > 16): 16 + (1 if A is applied in sparse form) + (2 if H is) + (4 if the passes run as one persistent
   kernel, TGP_OPT_DENSE_FUSED) */
int tgp_kernel_variant(const tgp_handle* h);
/* number of calls served by replaying a recorded hipGraph since the handle was created (TGP_OPT_GRAPH; measurement / tests) */
int64_t tgp_graph_replays(const tgp_handle* h);
/* Diagnostics of TGP_OPT_STEADY: how many of the series' steps the last call ran with stationary gains, out of `total` = T * p --
   the stationary-gain engine: T - n0 (tgp_logpdf, tgp_[logpdf_and_]posterior_marginals); the general engine: the steps the forward pass of
   the last posterior-path call ran in the mean-only form. */
int tgp_steady_steps(tgp_handle* h, int64_t* mean_only, int64_t* total);
/* Diagnostics of TGP_OPT_SWEEP for the last tgp_logpdf / tgp_[logpdf_and_]posterior_marginals call. info [8]: served by the sweep engine (0 / 1),
   steps per chunk, forward warm-up, backward warm-up, waves, attempts (launches), status bits of the last attempt (1 forward / 2 backward warm-up
   too short, 4 not positive definite, 8 non-finite), state for the bound model (0 untried, 1 serves, -1 declined).  dist [2]: the largest
   relative distance between a warm-up's end state and the run that reproduces it, forwards / backwards (the checks' 1e-12). Either may be NULL. */
int tgp_sweep_info(tgp_handle* h, int64_t* info, double* dist);

/* ---- model: replaces the LGSSM / GaussMarkovModel containers -----------------------------------
 * lgssm.jl:9-12, gauss_markov_model.jl:20-32 (As, as, Qs, x0) + emissions (A = H', a = h, Q = R).
 * x0m (d) and x0P (d*d) are always host pointers.
 * d <= 16 binds the time-parallel scan engine (p <= 64, diagonal noise). d > 16 (the ArrayStorage-sized models of
 * space_time/to_gauss_markov.jl:1-20; p <= 256) binds the dense engine -- one fp64-MFMA kernel chain per time step -- which
 * serves tgp_logpdf, tgp_filter, tgp_posterior (Forward priors), tgp_[logpdf_and_]posterior_marginals, tgp_marginals and
 * tgp_rand; the gradient, time-sharding and *_at entry points return TGP_EUNSUPPORTED there.
 * TOLERANCE OF THE LARGE-OUTPUT / BOTTLENECK EMISSIONS: the host side folds LargeOutputLGC and BottleneckLGC
 * (lgc.jl:179-204, :320-336) into ONE SmallOutput emission before it reaches this entry point -- an algebraic equivalence, not a restatement:
 * the reference's jitters on the predicted covariance (1e-10, lgc.jl:183) and on the bottleneck's projected noise (1e-12, lgc.jl:308-312)
 * are not carried through the p scalar updates.  Results agree with the reference's literal recursion to 1e-6 relative (the tolerance of the
 * reference's own tests for these types, test/models/linear_gaussian_conditionals.jl), against 1e-8 / 1e-10 for every other emission type. */
int tgp_model_set(tgp_handle* h, int64_t T, int d, int p, int ordering, uint32_t flags, const double* A,
                  const double* a, const double* Q, const double* H, const double* hh, const double* R,
                  const double* x0m, const double* x0P);
/* ---- model described by its LTI SDE + time stamps (irregular spacing): replaces broadcast_components for
 *      AbstractVector inputs, src/gp/lti_sde.jl:135-146 -- A_k = exp(F dt_k), Q_k = x0P - A_k x0P A_k' with
 *      dt_1 := 1 -- computed ON THE DEVICE into the tiled layout (no T host-side matrix exponentials, no [T][d*d]
 *      arrays). F [d*d] column-major, a [d] (shared), H / hh / R as in tgp_model_set (flags say shared or per-step),
 *      times [T] non-decreasing (checked: TGP_EINVAL otherwise; equal stamps are a step with A = I, Q = 0), x0P doubles as the
 *      stationary covariance. p == 1. All host pointers.
 *      A1, Q1 [d*d] (nullable): the FIRST transition. The reference's dt_1 := 1 is taken in each sub-kernel's own
 *      (stretched) time (lti_sde.jl:139 under :350-373,:404-418), so kernel algebra with ScaleTransforms has no
 *      single dt_1; the host evaluates that one block with the reference's rule and passes it here. */
int tgp_model_set_sde(tgp_handle* h, int64_t T, int d, int ordering, uint32_t flags, const double* F, const double* a,
                      const double* H, const double* hh, const double* R, const double* times, const double* A1,
                      const double* Q1, const double* x0m, const double* x0P);
/* replace only x0 (used for the carry-in state of a time shard) */
int tgp_model_set_x0(tgp_handle* h, const double* x0m, const double* x0P);

/* ---- logpdf(model::LGSSM, y): lgssm.jl:147-165 (+ missings.jl:8-13 when `missing` != NULL) ------ */
int tgp_logpdf(tgp_handle* h, const double* y, const uint8_t* missing, uint32_t flags, double* out);

/* ---- logpdf and its gradient w.r.t. `nparams` hyper-parameters, by forward-mode tangent scans (one pass of the
 *      dual-number instantiation of the engine per parameter). The reference has no gradient code: Mooncake.jl
 *      differentiates the generic Julia loop (test/gp/lti_sde.jl:203-206, bench/single_output_gps.jl:149-156).
 *      Forward models with every block shared (Fill: regular spacing, homoscedastic noise), p == 1.
 *      Tangents of the shared blocks per parameter k (host pointers): dA [np][d*d] column-major, da [np][d],
 *      dQ [np][d*d], dH [np][d], dh [np], dR [np], dx0m [np][d], dx0P [np][d*d]. grad_out [np] (host). */
int tgp_logpdf_grad(tgp_handle* h, const double* y, const uint8_t* missing, uint32_t flags, int nparams,
                    const double* dA, const double* da, const double* dQ, const double* dH, const double* dh,
                    const double* dR, const double* dx0m, const double* dx0P, double* lml_out, double* grad_out);

/* ---- the same for a model described by its SDE (tgp_model_set_sde: irregular spacing), d <= 4: the tangents of the per-step
 *      A_k = exp(F dt_k), Q_k = Pinf - A_k Pinf A_k' are built ON THE DEVICE from the tangents of F and Pinf -- dA_k by a central
 *      difference of the in-register exponential (relative step rel_step, 0 => 1e-6), dQ_k by the product rule -- into a
 *      tangent copy of the tiled transition record; the dual-number scan then runs in the general layout.
 *      dF, dPinf [nparams][d*d]; dA1, dQ1 [nparams][d*d] tangents of the explicit first transition (NULL if the model has
 *      none); da, dH [nparams][d]; dh, dR [nparams] (dR must be 0 for a per-step noise variance); dx0m, dx0P as above. */
int tgp_logpdf_grad_sde(tgp_handle* h, const double* y, const uint8_t* missing, uint32_t flags, int nparams, const double* dF,
                        const double* dPinf, const double* dA1, const double* dQ1, const double* da, const double* dH,
                        const double* dh, const double* dR, const double* dx0m, const double* dx0P, double rel_step,
                        double* lml_out, double* grad_out);

/* ---- logpdf and its gradient with respect to the MODEL BLOCKS by one adjoint (reverse-time) pass -------------------
 * Forward LTI models (every block shared: the reference's Fill layout of RegularSpacing inputs, lti_sde.jl:148-160) with
 * one noise variance, scalar observations, no missing data, d <= 8 -- the models of the stationary-gain engine. One
 * forward and one backward mean recursion over the series (the cost of a posterior-marginals call, whatever the number
 * of hyper-parameters) leave the sums behind d logpdf / d (A, a, Q, H, h, R, x0m, x0P); the host finishes with the head's
 * steps and a reverse sweep through the ~n0 steps of the covariance recursion. The caller contracts the block gradients
 * with d block / d theta of its own parametrisation (the O(1) host map lti_sde.jl:148-160). Any output may be NULL.
 * gA, gQ, gx0P [d*d] column-major (gQ, gx0P symmetrised), ga, gH, gx0m [d], ghh, gR [1]; all host pointers.
 * Replaces Mooncake's reverse mode over the sequential loop (bench/single_output_gps.jl:149-156, test/gp/lti_sde.jl:203-206).
 * d <= 6 (TGP_OPT_STEADY = 3, the default): plan and head on the host, ONE kernel behind the head.
 * TGP_EUNSUPPORTED when the engine does not apply (use tgp_logpdf_grad). */
int tgp_logpdf_adjoint(tgp_handle* h, const double* y, uint32_t flags, double* lml_out, double* gA, double* ga,
                       double* gQ, double* gH, double* ghh, double* gR, double* gx0m, double* gx0P);
/* The plan of the stationary-gain engine's ONE-LAUNCH path (TGP_OPT_STEADY = 3, the default), a pure host function: what
 * tgp_logpdf / tgp_[logpdf_and_]posterior_marginals build inside every call of a Forward LTI model with one noise variance and scalar
 * observations (d <= 8) before their single kernel -- the covariance half of lgssm.jl:99-238 (predict lgc.jl:46-52, update lgc.jl:247-257,
 * invert_dynamics lgssm.jl:231-238) run to its fixed point on the host, and the stationary mean recursions in modal form.
 * Model blocks as for tgp_model_set (host pointers, column-major, every block shared). Outputs (host):
 *   info_i [8]: why (0 applies, 1 covariance not settled within 623 steps, 2 not positive definite, 3 series shorter than head + tail,
 *               4 closed loop ill-conditioned in modal form, 5 mixes too slowly for a 1536-step halo, 6 smoother transient > 2048 steps,
 *               7 eigenvalue iteration failed), n0, n1, head steps, halo steps, complex pairs, waves per workgroup, 0
 *   info_d [4]: condition numbers of the two eigenvector matrices, spectral radius, residual of the rejected check
 *   modal_out [270] (may be NULL; 8-strided arrays): fd fo fb fa fw | gd go gc gw | M^8 re, im fwd, bwd | M^512 re, im fwd, bwd | WJ [8][8] |
 *               WG [8][8] | hh, R/S, 1/S, log S, sum of log S over the head, smoothed stationary variance
 *   tables_out (may be NULL; capacity 624 (d d + 2 d + 3) + 2048 + 80): h [8], modal mu0 [8], W [64], then per head step t <= n0:
 *               V^-1 (A K_t - A K) [d], 1/S_t, R/S_t, G_t [d d], c_t [d], smoothed variance, and the n1 tail variances. */
int tgp_steady_plan(int d, const double* A, const double* a, const double* Q, const double* H, const double* hh, const double* R,
                    const double* x0m, const double* x0P, int64_t T, int32_t* info_i, double* info_d, double* modal_out,
                    double* tables_out);

/* The host plan of the wide-state engine (8 < d <= 63: csrc/tgp_wide.hip) as a pure host function -- no handle, no GPU: the covariance half of
 * lgssm.jl:99-165 to its fixed point.  info [8]: why (0 = applies), n0 (head steps), halo, why_post (-1: not asked), n1 (steps at the series' end whose
 * smoothed variance is in its transient), halo_back, chunks, chunk length.  Kss [d], Sss: the stationary gain and innovation variance;
 * var_parts [2] (want_posterior): the stationary smoothed emission variance is var_parts[0] - var_parts[1] (+ the new noise). */
int tgp_wide_plan(int d, const double* A, const double* a, const double* Q, const double* H, const double* hh, const double* R, const double* x0m,
                  const double* x0P, int64_t T, int want_posterior, int64_t* info, double* Kss, double* Sss, double* var_parts);
/* ---- time segments of ONE series on the one-launch path (TGP_OPT_STEADY = 3): what a rank of a multi-GPU run calls (tgp_multi_* and the
 * one-process-per-GPU driver use them).  Both mean recursions of the stationary region forget a state within `halo` steps, so a segment needs
 * nothing of its neighbours but their `halo` observations next to the boundary: ONE all-gather of 2 halo observations per rank before the
 * call, one sum of the ranks' shares of the log marginal likelihood after it; no exchange of filter elements, no carry between ranks
 * (the reference has no counterpart: src/util/scan.jl:15-28 is one sequential loop).
 * tgp_segment_plan: host only. bounds [nseg + 1] (bounds[0] = 0, bounds[nseg] = T_total; interior ones multiples of 16). *applies = 1 when the
 *   one-launch path serves EVERY segment of this model and series (the plan is a function of the model blocks and T_total alone: every rank
 *   computes the same answer without communication), *halo = the number of neighbour observations a segment needs on each side.
 * tgp_segment_logpdf_and_posterior_marginals: the handle is bound to the segment's model (T = seg_hi - seg_lo, every block shared, the
 *   SERIES' x0); y_seg [T] the segment's observations, y_left [halo] those in front of it (NULL for the first segment), y_right [halo] those
 *   behind it (NULL for the last) -- ALWAYS device pointers (TGP_IN_DEVICE says where Rnew lives, TGP_OUT_DEVICE where mean / var go).
 *   mean_out / var_out [T] (NULL: logpdf only) as in tgp_posterior_marginals. *lml_share: this segment's share; the shares of all segments
 *   add up to logpdf(model, y) of lgssm.jl:147-165. A NaN observation makes the share NaN (callers fall back to the general protocol). */
int tgp_segment_plan(tgp_handle* h, int64_t T_total, int nseg, const int64_t* bounds, int32_t* applies, int32_t* halo);
int tgp_segment_logpdf_and_posterior_marginals(tgp_handle* h, int64_t T_total, int64_t seg_lo, int64_t seg_hi, const double* y_seg,
                                               const double* y_left, const double* y_right, const double* Rnew, uint32_t flags,
                                               double* mean_out, double* var_out, double* lml_share);
/* the host half of it, a pure host function (tests; callers that keep the device record): rec = tgp_adjoint_record_size(d)
 * doubles as tgp_steady.hpp lays them out, y_head = the first n_head observations (n_head >= 512 * head tiles) */
int tgp_adjoint_record_size(int d);
int tgp_adjoint_finish(int d, const double* rec, const double* y_head, int64_t n_head, double* gA, double* ga, double* gQ,
                       double* gH, double* ghh, double* gR, double* gx0m, double* gx0P);

/* ---- _filter(model, y): lgssm.jl:171-187. m_out [T][d], P_out [T][d*d] (either may be NULL);
 *      lml_out (host, may be NULL) receives the log marginal likelihood as a by-product.
 *      A Forward LTI model with scalar observations, one noise variance, no missing data and d <= 8 runs its head on the host and
 *      everything behind it as ONE kernel (TGP_OPT_STEADY = 3, the default). */
int tgp_filter(tgp_handle* h, const double* y, const uint8_t* missing, uint32_t flags, double* m_out,
               double* P_out, double* lml_out);

/* ---- posterior(model, y): lgssm.jl:193-238. Materialises the time-reversed model:
 *      G [T][d*d], g [T][d], L [T][d*d] (all three or none), xfm (d) / xfP (d*d): x0 of the posterior
 *      (host pointers). Forward priors (step_posterior(::Forward), :215-221) and Reverse priors (step_posterior(::Reverse), :223-228:
 *      invert_dynamics(xp, xf, t) as the reference calls it, x0 = the state after the last step's predict; G, g, L must be requested).
 *      A Forward LTI model with scalar observations, one noise variance, no missing data and d <= 8: the head's transitions on the host,
 *      everything behind it by the filter's ONE kernel (G, L constant there; TGP_OPT_STEADY = 3, the default). */
int tgp_posterior(tgp_handle* h, const double* y, const uint8_t* missing, uint32_t flags, double* G,
                  double* g, double* L, double* xfm, double* xfP);

/* ---- marginals(replace_observation_noise_cov(posterior(model, y), Rnew)): the benchmarked
 *      `marginals(posterior(fx, y)(x))` path, posterior_lti_sde.jl:27-36 -> lgssm.jl:99-115,193-238,
 *      missings.jl:35-41. Forward filter + RTS smoother + emission predict, nothing materialised.
 *      Rnew [T] (or one scalar with TGP_SHARED_R); mean_out, var_out [T]; lml_out host, may be NULL. */
int tgp_posterior_marginals(tgp_handle* h, const double* y, const uint8_t* missing, const double* Rnew,
                            uint32_t flags, double* mean_out, double* var_out, double* lml_out);

/* ---- logpdf(model, y) AND marginals(replace_observation_noise_cov(posterior(model, y), Rnew)) of the same (model, y) in ONE
 *      forward filter + RTS smoother: the log marginal likelihood is a by-product of the filter that the posterior needs anyway
 *      (lgssm.jl:147-165 and :193-238 run the same predict / posterior_and_lml recursion), so a caller that wants both -- model
 *      fitting followed by prediction at the training inputs, the benchmarked pair -- pays for one pass 1 / pass 2, not two. */
int tgp_logpdf_and_posterior_marginals(tgp_handle* h, const double* y, const uint8_t* missing, const double* Rnew, uint32_t flags,
                                       double* lml_out, double* mean_out, double* var_out);

/* ---- posterior marginals through an ALTERNATIVE emission block: N(Hn x_t + hn, Hn P_t Hn' + Rn) under the smoothed
 *      state, pn functionals per time step that are not the model's observations -- what the reference gets by swapping
 *      the emissions of the posterior model (space_time/pseudo_point.jl:198-235 approx_posterior_marginals, and
 *      posterior_lti_sde.jl's prediction at the training inputs with other outputs) without materialising that model.
 *      Hn [pn][d], hn [pn] host; Rn [T][pn] (or [pn] with TGP_SHARED_R); mean_out, var_out [T][pn].
 *      Forward LTI models served by the group-per-chunk smoother (d = 5..16); TGP_EUNSUPPORTED otherwise. */
int tgp_posterior_marginals_at(tgp_handle* h, const double* y, const uint8_t* missing, int pn, const double* Hn,
                               const double* hn, const double* Rn, uint32_t flags, double* mean_out, double* var_out,
                               double* lml_out);

/* ---- marginals(model): lgssm.jl:99-115 for the model as given (prior marginals for a Forward prior,
 *      smoothing marginals for a materialised Reverse posterior). */
int tgp_marginals(tgp_handle* h, uint32_t flags, double* mean_out, double* var_out);

/* ---- rand(rng, replace_observation_noise_cov(posterior(model, y), Rnew)) with the randomness supplied
 *      (posterior_lti_sde.jl:48-58 -> lgssm.jl:65-91 on the reverse-time model of lgssm.jl:193-221) WITHOUT
 *      evaluating that model (T x (2 d^2 + d) doubles): Forward LTI models with scalar observations, one noise
 *      variance, no missing data and d <= 6 run the filter and the reverse-time draw in ONE kernel over
 *      y and the draws. eps_t [T][d], eps_e [T] where TGP_IN_DEVICE says (as y and Rnew),
 *      eps_0 [d] host; eps_t[t] / eps_e[t] drive the transition / emission of step t and eps_0 the draw of the
 *      final filtering state, exactly as tgp_rand on the evaluated posterior uses them. Rnew: one value
 *      (TGP_SHARED_R) or T. y_out [T]. TGP_EUNSUPPORTED: not a model of this path -- evaluate the posterior
 *      (tgp_posterior), bind it as a Reverse model and call tgp_rand (what the Python mirror does). */
int tgp_posterior_rand(tgp_handle* h, const double* y, const double* Rnew, const double* eps_t, const double* eps_e,
                       const double* eps_0, uint32_t flags, double* y_out);

/* ---- logpdf(replace_observation_noise_cov(posterior(model, y), R_new), y_new) (posterior_lti_sde.jl:62-78 ->
 *      lgssm.jl:147-151 on the reverse-time model of lgssm.jl:193-221) without a posterior: two observations of
 *      one latent value with independent noise are one observation of it,
 *          N(y; f, R) N(y_new; f, R_new) = N(ybar; f, Rbar) N(y - y_new; 0, R + R_new),
 *          Rbar = R R_new / (R + R_new),  ybar = (R_new y + R y_new) / (R + R_new),
 *      so the posterior's logpdf of a Forward model with scalar observations is
 *          tgp_logpdf(model with Rbar, ybar) + pair - tgp_logpdf(model, y)
 *      -- two calls on whatever engine the prior has. This entry point is the one pass over the two
 *      series that forms ybar, Rbar and pair = sum_t log N(y_t - y_new_t; 0, R_t + R_new_t) over the steps observed on
 *      both sides. ALL series pointers are DEVICE pointers on h's device (n doubles / n bytes): y, y_new, the masks
 *      (optional; 1 = missing), ybar, Rbar, missing_bar. R, R_new: nR / nR_new = 1 (one variance, host or device
 *      pointer) or n (per step, device). A step missing on one side keeps the other side's observation and variance;
 *      on both sides it is missing in the result (missing_bar, needed only with two masks). Rbar may be NULL when both
 *      variances are single numbers and nothing is missing (Rbar = R R_new / (R + R_new) for the caller to form).
 *      pair_out: host. h needs no model; its stream orders the pass. */
int tgp_pair_statistic(tgp_handle* h, int64_t n, const double* y, const uint8_t* missing, const double* R, int64_t nR,
                       const double* y_new, const uint8_t* missing_new, const double* R_new, int64_t nR_new, double* ybar,
                       double* Rbar, uint8_t* missing_bar, double* pair_out);

/* ---- tgp_logpdf of the bound model with ANOTHER noise variance R (one number; every other block as bound): the joint
 *      model of the identity above is the prior with Rbar, and replace_observation_noise_cov (missings.jl:35-41) of an
 *      LTI model changes nothing else -- no second model bound. Served by the one-launch paths only (their plans are host
 *      functions of the blocks, rebuilt per call): Forward LTI models with scalar observations, no missing data;
 *      TGP_EUNSUPPORTED otherwise (bind the model with the new variance and call tgp_logpdf). y as TGP_IN_DEVICE says. */
int tgp_logpdf_noise(tgp_handle* h, const double* y, uint32_t flags, double R, double* out);

/* ---- rand(rng, model) with the randomness supplied: lgssm.jl:65-91, lgc.jl:84-87,241-243,
 *      gaussian.jl:35-43. eps_t [T][d], eps_e [T], eps_0 [d] (eps_0 always host). y_out [T].
 *      A Forward LTI model (every block shared) with scalar observations and d <= 8 runs as ONE kernel over the draws
 *      (TGP_OPT_STEADY = 3, the default); everything else on the general engine's affine scan. */
int tgp_rand(tgp_handle* h, const double* eps_t, const double* eps_e, const double* eps_0, uint32_t flags,
             double* y_out);

/* ---- time sharding across GPUs -------------------------------------------
 * A shard reduces its whole segment to ONE scan element, the host exchanges the (tiny) elements
 * (torch.distributed all_gather over RCCL), folds those of the shards to its left onto x0, sets the
 * result with tgp_model_set_x0 and finishes with the normal entry points + TGP_REUSE_REDUCE.
 *   kind 0: filter element  (d*d + d + d(d+1)/2 + d + d(d+1)/2 doubles, packed)
 *   kind 1: smoother element (d*d + d + d(d+1)/2 doubles) -- valid after tgp_smoother_forward        */
int tgp_elem_size(int kind, int d);
int tgp_segment_reduce(tgp_handle* h, const double* y, const uint8_t* missing, uint32_t flags, double* elem_out);
/* host-side monoid operations on packed elements / states (m (d), P (d*d) column-major) */
int tgp_elem_apply(int kind, int d, const double* elem, const double* m, const double* P, double* m_out, double* P_out);
int tgp_elem_combine(int kind, int d, const double* earlier, const double* later, double* out);

/* two-phase posterior marginals for shards: forward (filter + reverse chunk elements; returns the
 * segment's smoother element and its final filtered state), then backward from the smoothed state at
 * the segment end (xs_m/xs_P host; NULL => the segment's own final filtered state). */
int tgp_smoother_forward(tgp_handle* h, const double* y, const uint8_t* missing, uint32_t flags,
                         double* rev_elem_out, double* xfm, double* xfP, double* lml_out);
int tgp_smoother_backward(tgp_handle* h, const double* xs_m, const double* xs_P, const double* Rnew,
                          uint32_t flags, double* mean_out, double* var_out);

/* ---- device-resident exchange for shards ----------------------------------------------------------
 * The same protocol with NO host round trip between the phases: every call below only enqueues work on
 * the handle's stream (tgp_set_stream: use the stream the caller's collectives are ordered against).
 * `slot_dev` / `gathered_dev` / `stats_dev` are DEVICE buffers owned by the caller; `gathered_dev` is
 * what an all-gather of one slot per rank leaves (rank-major, world * tgp_shard_slot_size doubles).
 *   tgp_shard_reduce            pass 1 of this segment -> one filter element in slot_dev  (phase-0 slot)
 *   [caller: all-gather phase-0 slots]
 *   tgp_shard_fold              carry-in of this segment = elements 0..rank-1 folded onto the prior x0 (k_fold)
 *   tgp_shard_logpdf            pass 2 -> stats_dev[0..3] = (lml, n missing, not-PD count, Cholesky flag)
 *   [caller: all-reduce(sum) of stats, ONE device-to-host copy]
 * or, for posterior marginals,
 *   tgp_shard_smoother_forward  pass 2 + reverse elements -> (smoother element | final filtered state) (phase-1 slot)
 *   [caller: all-gather phase-1 slots]
 *   tgp_shard_smoother_backward folds ranks W-1..rank+1 onto the last rank's final state, smooths this segment;
 *                               the ONE synchronisation of the whole call, reports the error flags of all phases */
int tgp_shard_slot_size(int phase, int d);
int tgp_shard_reduce(tgp_handle* h, const double* y, const uint8_t* missing, uint32_t flags, double* slot_dev);
int tgp_shard_fold(tgp_handle* h, const double* gathered_dev, int world, int rank);
int tgp_shard_logpdf(tgp_handle* h, double* stats_dev);
int tgp_shard_smoother_forward(tgp_handle* h, double* slot_dev);
int tgp_shard_smoother_backward(tgp_handle* h, const double* gathered_dev, int world, int rank, const double* Rnew,
                                uint32_t flags, double* mean_out, double* var_out, double* lml_out);

/* ---- time shards of the stationary-gain engine (LTI models of tgp_logpdf_adjoint's class; csrc/tgp_steady.hpp) ----------------------
 * Two halves around ONE all-gather. begin: covariance set-up, pass 1 and a provisional carry pass of this segment; leaves the segment's
 * element (mu behind it, lam in front of it, Phi^L, G^L, B_L, an "applies" word) in slot_dev (tgp_shard_steady_slot_size(d) doubles).
 * finish: the chain of gathered elements -> this segment's real boundary, carry pass, pass 2 (posterior marginals when begin was
 * called with posterior != 0), reduction; *served = 0 when ANY rank found that the engine does not apply to its segment (covariance
 * not settled, segment shorter than head / tail, a segment that hands its end on but is not a whole number of 512-step tiles):
 * outputs are then undefined and the caller runs the tgp_shard_* protocol above. first / last: the segment starts / ends the series.
 * lml_out: this segment's share of the log marginal likelihood. Only enqueues until finish's one synchronisation. */
int tgp_shard_steady_slot_size(int d);
int tgp_shard_steady_begin(tgp_handle* h, const double* y, uint32_t flags, int first, int last, int posterior, double* slot_dev);
int tgp_shard_steady_finish(tgp_handle* h, const double* gathered_dev, int world, int rank, const double* Rnew, uint32_t flags,
                            double* mean_out, double* var_out, double* lml_out, int* served);

/* ---- multi-GPU handle: the same protocol inside ONE process (SURVEY.md 8b "Threading", 8e) -----------
 * tgp_create_multi owns, per listed device, one tgp_handle with its own HIP stream, one host worker thread and
 * one RCCL communicator (ncclCommInitAll; librccl is opened at run time). Rank r serves the contiguous time
 * segment [t0_r, t1_r) of tgp_multi_segment. A call runs the tgp_shard_* phases of every rank concurrently
 * and exchanges the per-segment scan elements with ncclAllGather on the ranks' streams -- twice per posterior
 * call, once per logpdf; the W x 4 result words are summed on the host (one process: no all-reduce needed).
 * No bulk data crosses xGMI; inputs and outputs stay sharded: every per-call array argument is an ARRAY OF
 * ndev POINTERS, entry r addressing rank r's segment (host memory: `y + t0_r`; or, with TGP_IN_DEVICE /
 * TGP_OUT_DEVICE, memory of rank r's device). `missing` may be NULL; with TGP_SHARED_R each Rnew[r] points to
 * one value. devices == NULL: devices 0..ndev-1 (ndev == 0: every visible device). A device listed more than
 * once (several ranks on one GPU: tests) or TGP_MULTI_TRANSPORT=copy replaces RCCL by event-ordered peer copies.
 * (Tests only: TGP_MULTI_RCCL_LIB names another library with librccl's four entry points, TGP_MULTI_TRANSPORT=rccl asks for the
 * RCCL branch even when ranks share a device -- tests/stub_rccl.cpp.)
 * Scan engine only (d <= 16, Forward ordering); the dense path (d > 16) does not time-shard.
 * Replaces: the sequential loop of src/util/scan.jl:15-28 behind logpdf (lgssm.jl:147-151) and
 * marginals(posterior(...)) (lgssm.jl:193-200, :111-115) for a series that spans the GPUs of a node. */
typedef struct tgp_multi tgp_multi;
int tgp_create_multi(tgp_multi** m, int ndev, const int* devices);
int tgp_destroy_multi(tgp_multi* m);
const char* tgp_multi_last_error(const tgp_multi* m);
int tgp_multi_ndev(const tgp_multi* m);
const char* tgp_multi_transport(const tgp_multi* m); /* "rccl" or "copy (<why>)" */
tgp_handle* tgp_multi_handle(tgp_multi* m, int rank); /* borrowed: options, profile, diagnostics of one rank */
int tgp_multi_segment(int64_t T, int ndev, int rank, int64_t* t0, int64_t* t1);
int tgp_multi_set_option(tgp_multi* m, int option, int64_t value);
/* as tgp_model_set, host pointers for the WHOLE series: shared (Fill) blocks go to every rank, per-step arrays are
 * sliced per segment; x0 is the prior of the whole series */
int tgp_multi_model_set(tgp_multi* m, int64_t T, int d, int p, int ordering, uint32_t flags, const double* A,
                        const double* a, const double* Q, const double* H, const double* hh, const double* R,
                        const double* x0m, const double* x0P);
int tgp_multi_logpdf(tgp_multi* m, const double* const* y, const uint8_t* const* missing, uint32_t flags, double* out);
int tgp_multi_posterior_marginals(tgp_multi* m, const double* const* y, const uint8_t* const* missing,
                                  const double* const* Rnew, uint32_t flags, double* const* mean_out,
                                  double* const* var_out);
int tgp_multi_logpdf_and_posterior_marginals(tgp_multi* m, const double* const* y, const uint8_t* const* missing,
                                             const double* const* Rnew, uint32_t flags, double* lml_out,
                                             double* const* mean_out, double* const* var_out);

/* ---- timing ------------------------------------------------------------------------------------ */
/* device time of the last call (hipEvent, kernels only) and its host<->device copy times, ms; needs TGP_OPT_TIMING = 1 */
int tgp_last_timing(const tgp_handle* h, double* kernel_ms, double* h2d_ms, double* d2h_ms);
/* per-kernel hipEvent profile accumulated since tgp_profile_reset (TGP_OPT_PROFILE = 1) */
int tgp_profile_reset(tgp_handle* h);
int tgp_profile_count(tgp_handle* h);
int tgp_profile_get(tgp_handle* h, int idx, char* name, int name_cap, double* total_ms, int64_t* calls);
/* one EMPTY kernel inside the same hipEvent bracket (entry "k_empty"): what the bracket itself reads on this stack (about 6 us on ROCm 7.2 / MI355X --
 * the events' own packets), i.e. how far a hipEvent duration stands above the duration rocprofv3 --kernel-trace reports for the same kernel */
int tgp_profile_empty_launch(tgp_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* TGP_HIP_H */
