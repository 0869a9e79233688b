/* ORACLE (test infrastructure only -- never linked into the product library).
 * C restatement of the reference's sequential scalar-output Kalman recursions with compile-time
 * state dimension (the analogue of the reference's SArrayStorage path: fully unrolled fixed-size
 * stack matrices). Used (a) as the large-T checker for the HIP path and (b) by bench.py's
 * `cpu_baseline` leg (kind "port", 1 core -- the reference's scan is single-threaded,
 * /root/reference/src/util/scan.jl:15-28). See seq_kalman_body.inc for the per-function citations.
 * PARITY UNPINNED vs reference-run outputs (no Julia in this image); checked against
 * oracle/lgssm_ref.py (tests/test_oracle_c.py), which the reference's identities pin.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define LOG2PI 1.8378770664093454835606594728112
#define CAT_(a, b) a##_d##b
#define CAT(a, b) CAT_(a, b)
#define NAME(f) CAT(f, D)

#define D 1
#include "seq_kalman_body.inc"
#undef D
#define D 2
#include "seq_kalman_body.inc"
#undef D
#define D 3
#include "seq_kalman_body.inc"
#undef D
#define D 4
#include "seq_kalman_body.inc"
#undef D
#define D 5
#include "seq_kalman_body.inc"
#undef D
#define D 6
#include "seq_kalman_body.inc"
#undef D
#define D 7
#include "seq_kalman_body.inc"
#undef D
#define D 8
#include "seq_kalman_body.inc"
#undef D

#define DISPATCH(fn, ...)                                   \
    switch (d) {                                            \
        case 1: return fn##_d1(__VA_ARGS__);                \
        case 2: return fn##_d2(__VA_ARGS__);                \
        case 3: return fn##_d3(__VA_ARGS__);                \
        case 4: return fn##_d4(__VA_ARGS__);                \
        case 5: return fn##_d5(__VA_ARGS__);                \
        case 6: return fn##_d6(__VA_ARGS__);                \
        case 7: return fn##_d7(__VA_ARGS__);                \
        case 8: return fn##_d8(__VA_ARGS__);                \
        default: return 4;                                  \
    }

int oracle_seq_filter(int d, int64_t T, const double *A, int64_t sA, const double *a, int64_t sa,
                      const double *Q, int64_t sQ, const double *H, int64_t sH, const double *h, int64_t sh,
                      const double *R, int64_t sR, const double *y, const double *x0m, const double *x0P,
                      double *lml_out, double *m_out, double *P_out) {
    DISPATCH(seq_filter, T, A, sA, a, sa, Q, sQ, H, sH, h, sh, R, sR, y, x0m, x0P, lml_out, m_out, P_out)
}

int oracle_seq_posterior(int d, int64_t T, const double *A, int64_t sA, const double *a, int64_t sa,
                         const double *Q, int64_t sQ, const double *H, int64_t sH, const double *h, int64_t sh,
                         const double *R, int64_t sR, const double *y, const double *x0m, const double *x0P,
                         double *G, double *g, double *L, double *xfm, double *xfP) {
    DISPATCH(seq_posterior, T, A, sA, a, sa, Q, sQ, H, sH, h, sh, R, sR, y, x0m, x0P, G, g, L, xfm, xfP)
}

int oracle_seq_posterior_marginals(int d, int64_t T, const double *A, int64_t sA, const double *a, int64_t sa,
                                   const double *Q, int64_t sQ, const double *H, int64_t sH, const double *h,
                                   int64_t sh, const double *R, int64_t sR, const double *y,
                                   const double *x0m, const double *x0P, const double *Rnew, int64_t sRn,
                                   double *G, double *g, double *L, double *mean_out, double *var_out) {
    DISPATCH(seq_posterior_marginals, T, A, sA, a, sa, Q, sQ, H, sH, h, sh, R, sR, y, x0m, x0P, Rnew, sRn,
             G, g, L, mean_out, var_out)
}

int oracle_seq_prior_marginals(int d, int64_t T, const double *A, int64_t sA, const double *a, int64_t sa,
                               const double *Q, int64_t sQ, const double *H, int64_t sH, const double *h,
                               int64_t sh, const double *R, int64_t sR, const double *x0m, const double *x0P,
                               double *mean_out, double *var_out) {
    DISPATCH(seq_prior_marginals, T, A, sA, a, sa, Q, sQ, H, sH, h, sh, R, sR, x0m, x0P, mean_out, var_out)
}

int oracle_seq_rand(int d, int64_t T, const double *A, int64_t sA, const double *a, int64_t sa,
                    const double *Q, int64_t sQ, const double *H, int64_t sH, const double *h, int64_t sh,
                    const double *R, int64_t sR, const double *x0m, const double *x0P,
                    const double *eps_t, const double *eps_e, const double *eps_0, double *y_out) {
    DISPATCH(seq_rand, T, A, sA, a, sa, Q, sQ, H, sH, h, sh, R, sR, x0m, x0P, eps_t, eps_e, eps_0, y_out)
}
