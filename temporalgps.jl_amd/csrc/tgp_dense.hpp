// Dense large-state Kalman engine (state dimension d > 16): sequential in time, every time step a short fixed chain of
// fp64 MFMA (v_mfma_f64_16x16x4_f64) kernels. Host-side interface used by tgp_api.hip; kernels in tgp_dense.hip.
//
// Replaces, for ArrayStorage-sized models (the separable space-time path, BASELINE config 5: d = 768, p = 256):
//   predict                      /root/reference/src/models/linear_gaussian_conditionals.jl:46-52
//   posterior_and_lml(SmallOutputLGC)                                   .../linear_gaussian_conditionals.jl:129-151
//   step_logpdf / step_filter / step_marginals (Forward and Reverse)    /root/reference/src/models/lgssm.jl:99-187
//   invert_dynamics + the Reverse step_marginals (RTS smoother)         /root/reference/src/models/lgssm.jl:111-115,215-238
// for the dense A, Q, H that lgssm_components(::Separable, ...) builds (src/space_time/to_gauss_markov.jl:1-20).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

namespace tgp_dense {

struct Engine;

struct ModelDesc {
    int64_t T;
    int d, p, ordering;
    // device pointers in the ABI layout (tgp_hip.h): A, Q column-major d x d blocks, H [p][d] row-major, a [d], h [p], R [p];
    // stride 0 == shared block
    const double *A, *a, *Q, *H, *h, *R;
    int64_t sA, sa, sQ, sH, sh, sR;
    const double *x0m, *x0P;   // host
};

struct KernelTime {
    const char* name;
    double ms;
    int64_t calls;
};

Engine* create(int device);
void destroy(Engine* e);
const std::string& last_error(const Engine* e);
// options: profile (per-kernel hipEvents on a sampled subset of steps)
void set_profile(Engine* e, int on);
// 1 (default): a shared A / H with at most 8 entries per row is applied in sparse (ELL) form; 0: always the dense MFMA GEMMs
void set_structure(Engine* e, int on);
// posterior_marginals: length of the re-filtered segments (0 = automatic: all T steps stored if they fit, else ~sqrt(T))
void set_segment(Engine* e, int64_t steps);
// mid-sized states (d <= 64, p <= 16): 1 (default) the persistent single-kernel passes, 0 the per-step kernel chain
void set_fused(Engine* e, int on);
int structure(const Engine* e);   // bit 0: A sparse, bit 1: H sparse, bit 2: persistent single-kernel passes (current model)
int profile_count(Engine* e);
KernelTime profile_get(const Engine* e, int idx);
void profile_reset(Engine* e);

// (re)pack the model into the padded device layout the kernels read. Returns a TGP_* code.
int model_set(Engine* e, const ModelDesc& m, hipStream_t stream);
int set_x0(Engine* e, const double* x0m, const double* x0P, hipStream_t stream);

// y [T][p], mask [T][p] (nullable) device pointers. result8: device buffer of 8 doubles: [0] lml (incl. the missing-data
// volume compensation), [1] number of missing elements, [2] != 0 when a Cholesky met a non-positive pivot (first such step + 1).
// m_out [T][d] / P_out [T][d*d] (nullable, device): filtering distributions.
int filter(Engine* e, const double* y, const uint8_t* mask, double* m_out, double* P_out, double* result8, hipStream_t stream);

// marginals(replace_observation_noise_cov(posterior(model, y), Rnew)): forward filter storing the filtering states,
// then the RTS recursion. Rnew [T][p] (sRn = p) or [p] (sRn = 0), device. mean_out / var_out [T][p] device (diagonal).
int posterior_marginals(Engine* e, const double* y, const uint8_t* mask, const double* Rnew, int64_t sRn, double* mean_out,
                        double* var_out, double* result8, hipStream_t stream);

// marginals(model) of the model as given (prior marginals for a Forward model): diagonal of H P H' + R.
// posterior(prior, y) evaluated: per-step (G, g, L) (column-major d x d / d, device pointers; all three or none) and the final
// filtering state (host pointers, nullable)
int posterior(Engine* e, const double* y, const uint8_t* mask, double* G_out, double* g_out, double* L_out, double* xfm_host, double* xfP_host,
              double* result8, hipStream_t stream);
// rand with supplied noise: x0_host the drawn initial state (d, host), eps_t (T x d), eps_e (T x p), y_out (T x p) device pointers
int rand(Engine* e, const double* x0_host, const double* eps_t, const double* eps_e, int small_out, double* y_out, hipStream_t stream);
int marginals(Engine* e, double* mean_out, double* var_out, double* result8, hipStream_t stream);

}  // namespace tgp_dense
