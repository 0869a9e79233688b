// Small fixed-size fp64 linear algebra + the two scan monoids of the time-parallel Kalman engine.
// Everything is a template on the state dimension D with fully unrolled loops, so that on gfx950
// every matrix lives in VGPRs of ONE lane (one scan element / one chunk of steps per lane).
// Column-major d x d blocks throughout (the reference's Julia layout).
//
// Reference semantics restated here (file:line under /root/reference/src):
//   predict              models/linear_gaussian_conditionals.jl:46-52   (A*Symmetric(P))*A' + Q
//   posterior_and_lml    models/linear_gaussian_conditionals.jl:247-257 (ScalarOutputLGC)
//   invert_dynamics      models/lgssm.jl:231-238 (jitter 1e-10)
// The associative-scan elements (filter monoid (Abar,b,C,eta,J), affine monoid (E,g,L)) are NOT in
// the reference (its scan is a sequential loop, util/scan.jl:15-28); they follow Sarkka &
// Garcia-Fernandez, "Temporal Parallelization of Bayesian Smoothers" (IEEE TAC 2021), SURVEY.md 7.3.
#pragma once
#include <math.h>
#include <stdint.h>

// Compiler-level barrier for memory operations: loads issued before it are not sunk past it (software prefetch).
#if defined(__HIP_DEVICE_COMPILE__)
#define TGP_ISSUE_BARRIER() asm volatile("" ::: "memory")
#else
#define TGP_ISSUE_BARRIER() ((void)0)
#endif
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TGP_HD __host__ __device__ __forceinline__
#define TGP_LAMBDA_INLINE __attribute__((always_inline))
#else
#define TGP_HD inline
#define TGP_LAMBDA_INLINE
#endif

// d <= 8: fully unrolled (matrices in registers). The d = 9..16 translation units are compiled with TGP_NO_UNROLL: their
// matrices live in private memory anyway (out-of-line building blocks), and rolled loops keep code size and compile
// time sane.
#define TGP_NOUNROLL_LOOP _Pragma("nounroll")
#if defined(TGP_NO_UNROLL)
#define TGP_UNROLL _Pragma("nounroll")
#else
#define TGP_UNROLL _Pragma("unroll")
#endif

// Everything lives in namespace TGP_NS (default `tgp`). A translation unit may be compiled a second time under
// another namespace with another inlining policy (TGP_BIG_D): see tgp_inst_d5i.hip and the run-time variant check
// in tgp_api.hip.
#ifndef TGP_NS
#define TGP_NS tgp
#endif

// State dimensions >= TGP_BIG_D do not fit one lane's register file (d = 6: ~800 VGPRs of live matrices
// against 512): fully inlined, hipcc 7.2 spills hundreds of VGPRs and SGPRs and we measured silently wrong
// results from that path. For those D the per-step / per-combine building blocks are real (noinline)
// device functions: each has a small register footprint and the matrices live in the lane's private
// (scratch) memory by construction. d <= 4 stays fully inlined in registers.
#ifndef TGP_BIG_D
#define TGP_BIG_D 5
#endif
#if defined(__HIPCC__)
#define TGP_NOINLINE __host__ __device__ __attribute__((noinline))
#else
#define TGP_NOINLINE __attribute__((noinline))
#endif

namespace TGP_NS {

constexpr double kLog2Pi = 1.8378770664093454835606594728112;
constexpr double kLargeVar = 1e15;  // missings.jl:43

using ::fabs;
using ::fma;
using ::log;
using ::sqrt;

// ---------------------------------------------------------------- forward-mode dual number (value, one tangent)
// The whole engine (scan elements, combines, the per-step recursion) is instantiated a second time over Dual in
// namespace tgp::ad: d logpdf / d theta for ONE hyper-parameter per pass, exact to rounding, with the same
// parallel structure as the value computation ("tangent scans"). Control flow (pivoting,
// positivity checks) follows the value part.
struct Dual {
    double v, d;
    TGP_HD Dual() : v(0.0), d(0.0) {}
    TGP_HD Dual(double v_) : v(v_), d(0.0) {}
    TGP_HD Dual(double v_, double d_) : v(v_), d(d_) {}
};
TGP_HD Dual operator+(Dual a, Dual b) { return Dual(a.v + b.v, a.d + b.d); }
TGP_HD Dual operator-(Dual a, Dual b) { return Dual(a.v - b.v, a.d - b.d); }
TGP_HD Dual operator-(Dual a) { return Dual(-a.v, -a.d); }
TGP_HD Dual operator*(Dual a, Dual b) { return Dual(a.v * b.v, ::fma(a.v, b.d, a.d * b.v)); }
TGP_HD Dual operator/(Dual a, Dual b) {
    const double q = a.v / b.v;
    return Dual(q, (a.d - q * b.d) / b.v);
}
TGP_HD Dual& operator+=(Dual& a, Dual b) { a = a + b; return a; }
TGP_HD Dual& operator-=(Dual& a, Dual b) { a = a - b; return a; }
TGP_HD Dual& operator*=(Dual& a, Dual b) { a = a * b; return a; }
TGP_HD bool operator>(Dual a, Dual b) { return a.v > b.v; }
TGP_HD bool operator<(Dual a, Dual b) { return a.v < b.v; }
TGP_HD Dual fma(Dual a, Dual b, Dual c) { return Dual(::fma(a.v, b.v, c.v), ::fma(a.v, b.d, ::fma(a.d, b.v, c.d))); }
TGP_HD Dual sqrt(Dual a) {
    const double r = ::sqrt(a.v);
    return Dual(r, 0.5 * a.d / r);
}
TGP_HD Dual log(Dual a) { return Dual(::log(a.v), a.d / a.v); }
TGP_HD Dual fabs(Dual a) { return a.v < 0.0 ? -a : a; }
TGP_HD double value_of(double x) { return x; }
TGP_HD double value_of(Dual x) { return x.v; }
TGP_HD double tangent_of(double) { return 0.0; }
TGP_HD double tangent_of(Dual x) { return x.d; }

template <int D> struct Dim {
    static constexpr int DD = D * D;
    static constexpr int DS = D * (D + 1) / 2;       // packed symmetric
    static constexpr int NS = D + DS;                // state (m, P) packed
    static constexpr int NF = DD + D + DS + D + DS;  // filter element packed
    static constexpr int NA = DD + D + DS;           // affine element packed
};

using real_t = double;
#include "tgp_math_body.inc"

namespace ad {
using real_t = Dual;
#include "tgp_math_body.inc"
}  // namespace ad

}  // namespace TGP_NS
