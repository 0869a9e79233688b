# Reference-run fixtures: the ONLY route from "parity unpinned" to a pinned oracle (DESIGN 2; SURVEY 8c).
#
# The reference's own tests hold no golden vectors for the LGSSM hot path and this repository's build image has no Julia, so the oracle
# (oracle/lgssm_ref.py, oracle/seq_kalman.c) is pinned only by the identities the reference's tests check.  A maintainer with Julia runs
# THIS script once, in an environment that has TemporalGPs.jl (the reference, unmodified), AbstractGPs, KernelFunctions and NPZ:
#
#     julia --project=<env with TemporalGPs> julia/make_reference_fixtures.jl
#
# It writes tests/golden/reference_run.npz: for each case the inputs (y, drawn here with a fixed seed and stored) and the REFERENCE's
# outputs -- logpdf(fx, y) (src/models/lgssm.jl:147-165 through src/gp/lti_sde.jl:41-44), the posterior marginals at the same inputs
# (src/gp/posterior_lti_sde.jl:20-37 -> lgssm.jl:99-115,193-238) and the prior marginals (lti_sde.jl:33-35).  tests/test_reference_run.py
# loads the file when it is present and holds the oracle (CPU tier) and the device path (GPU tier) against it at the tolerances of every
# other parity test (logpdf 1e-10 relative, marginals 1e-8); commit the .npz with the script's output log.
#
# Cases = BASELINE.json's configurations at sizes the CPU reference finishes in seconds (cfg1 at its exact size).
using AbstractGPs, KernelFunctions, TemporalGPs, NPZ, Random, LinearAlgebra

const OUT = joinpath(@__DIR__, "..", "tests", "golden", "reference_run.npz")

stretched(k, s) = k ∘ ScaleTransform(s)

# name => (kernel, spec string the Python side parses, t0, dt, T, noise variance, storage)
const CASES = [
    ("cfg1_matern32_T10000", Matern32Kernel(), "matern32", 0.0, 0.1, 10_000, 0.1, SArrayStorage(Float64)),
    ("cfg2_matern52_T100000", Matern52Kernel(), "matern52", 0.0, 0.1, 100_000, 0.1, SArrayStorage(Float64)),
    ("cfg3_sum52_32_T50000", Matern52Kernel() + Matern32Kernel(), "sum(matern52,matern32)", 0.0, 0.1, 50_000, 0.1, SArrayStorage(Float64)),
    ("cfg3_sum52_52s_T50000", Matern52Kernel() + stretched(Matern52Kernel(), 2.0), "sum(matern52,stretched(2.0,matern52))", 0.0, 0.1, 50_000, 0.1,
     SArrayStorage(Float64)),
    ("cfg4_sum52_12_T200000", Matern52Kernel() + Matern12Kernel(), "sum(matern52,matern12)", 0.0, 0.1, 200_000, 0.1, SArrayStorage(Float64)),
    ("scaled_stretched_T20000", 1.7 * stretched(Matern52Kernel(), 0.6), "scaled(1.7,stretched(0.6,matern52))", -3.0, 0.05, 20_000, 0.3,
     ArrayStorage(Float64)),
    # wide states (round 6: csrc/tgp_wide.hip -- products of kernels, lti_sde.jl:377-400): d = 9 and d = 28
    ("wide_prod52_52_T30000", Matern52Kernel() * stretched(Matern52Kernel(), 0.7), "product(matern52,stretched(0.7,matern52))", 0.0, 0.1, 30_000, 0.1,
     ArrayStorage(Float64)),
    ("wide_periodic_x_matern32_T30000", ApproxPeriodicKernel() * Matern32Kernel(), "product(approx_periodic(7,1.0),matern32)", 0.0, 0.1, 30_000, 0.1,
     ArrayStorage(Float64)),
]

function main()
    out = Dict{String, Any}()
    names = String[]
    for (name, k, spec, t0, dt, T, s2, storage) in CASES
        rng = MersenneTwister(hash(name) % 2^31)
        f = to_sde(GP(k), storage)
        x = RegularSpacing(t0, dt, T)
        fx = f(x, s2)
        y = rand(rng, fx)
        lml = logpdf(fx, y)
        post = posterior(fx, y)
        pm = marginals(post(x))                       # the latent posterior at the training inputs (no observation noise)
        pr = marginals(fx)                            # prior marginals of the observations
        out[name * "/y"] = collect(Float64, y)
        out[name * "/logpdf"] = [Float64(lml)]
        out[name * "/post_mean"] = collect(Float64, mean.(pm))
        out[name * "/post_var"] = collect(Float64, var.(pm))
        out[name * "/prior_mean"] = collect(Float64, mean.(pr))
        out[name * "/prior_var"] = collect(Float64, var.(pr))
        out[name * "/meta"] = Float64[t0, dt, T, s2]
        push!(names, name * "|" * spec)
        println(rpad(name, 28), " T = ", T, "  logpdf = ", lml)
    end
    out["cases"] = join(names, ";")
    # (NPZ stores arrays: the case list travels as bytes)
    out["cases_bytes"] = Vector{UInt8}(out["cases"])
    delete!(out, "cases")
    npzwrite(OUT, out)
    println("wrote ", OUT)
end

main()
