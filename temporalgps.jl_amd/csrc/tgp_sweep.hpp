// The sweep engine (round 5): logpdf and posterior marginals of Forward models whose GAINS vary in time -- a missing-data mask,
// a noise variance per step, irregular spacing (transitions exp(F dt_k) in closed form) -- scalar observations, d <= 4.  What the
// reference's predict path produces (posterior_lti_sde.jl:20-37,97-131; missings.jl:25-41; lti_sde.jl:135-146) and what the stationary-gain
// engines (tgp_steady.hip, tgp_modal.hip) cannot serve: there the covariance half of the recursion is constant behind a head, here it
// depends on every step.
//
// One launch.  A lane owns a CHUNK of C consecutive steps and runs the reference's sequential recursion over it in registers (state
// (m, packed P): 9 doubles at d = 3 -- no scan elements, no carries between workgroups).  What a chunk needs from the steps before it is
// its start state, and the filter FORGETS: started from the stationary prior W steps early, it has the true state to rounding at the
// chunk's first step.  So every lane first warms up over the last W steps of its own chunk (the result is the NEXT lane's start state: one
// shift across the wave), then runs its chunk for real.  Backwards the same: the smoothing state at a chunk's end is the next lane's
// smoothing state at its first step, which that lane gets to rounding from a warm-up over its first Wb steps.  The filtering states a
// backward step needs are recomputed block by block from checkpoints (one state per B steps, written by the forward run) into LDS.
// A wave = 64 consecutive chunks, 62 of them its own (lane 0 only warms up for lane 1, lane 63 only for lane 62); waves are independent.
// Nothing is assumed about W: the run from the handed-over start must reproduce the warm-up's end state (and the same backwards) to
// `tol`, else the call reports it and the host repeats it with longer warm-ups (or hands it to the general engine).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "tgp_sweep_plan.hpp"

namespace tgp_sweep {

constexpr int64_t kMinT = 2048;      // shorter series stay on the general engine (a chunk must hold the warm-up)

struct Engine;
Engine* create();
void destroy(Engine* e);

struct Call {
    int64_t T = 0;
    const double* y = nullptr;            // device
    const uint8_t* mask = nullptr;        // device or null
    const double* R = nullptr;            // device per-step noise variance, or null (shared)
    const double* hh = nullptr;           // device per-step emission offset, or null (shared)
    const double* tau = nullptr;          // device [T] gaps (SDE)
    const double* Rnew = nullptr;         // device
    int rnew_per_step = 0;
    double* mean = nullptr;               // device; null: logpdf only
    double* var = nullptr;
};

// Chooses the geometry (chunk length, warm-ups) for this model and series; false: the engine declines (`why` says so).
// w_hint / wb_hint: warm-ups a previous call on the same bound model needed (0: estimate from the model).
bool plan(Engine* e, const ModelHost& m, int64_t T, int w_hint, int wb_hint, int num_cu, std::string* why);
int enqueue(Engine* e, hipStream_t stream, const Call& c, const char** kernel_name, std::string* err);
const char* kernel_name(int d, bool sde, bool post);
// After the stream has been synchronised.  status bits: 1 forward warm-up too short, 2 backward warm-up too short, 4 not positive definite,
// 8 non-finite values.  *w / *wb: the warm-ups the call ran with.
double finish(Engine* e, int* status, int* w, int* wb, double* dist_f, double* dist_b);
// geometry of the last plan (diagnostics / tests)
void geometry(const Engine* e, int* C, int* W, int* Wb, int64_t* nwaves);
// test hook: force the geometry of the next plan (0: automatic)
void force_geometry(Engine* e, int C, int W, int Wb);

}  // namespace tgp_sweep
