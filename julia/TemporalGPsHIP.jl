# TemporalGPsHIP.jl -- Julia-side glue for the MI355X backend (libtgp_hip.so, include/tgp_hip.h).
#
# NOT EXECUTED IN THIS REPOSITORY'S CI: neither the build image nor the GPU box has Julia
# (gpurun_out/julia.txt). It is the binding a TemporalGPs.jl maintainer adds; every ccall below has an
# executed twin in temporalgps.jl_amd/_lib.py + lgssm.py (ctypes), which is what the test-suite drives.
#
# Seam (reference, v0.7.3):
#   StorageType tag given to `to_sde`          src/util/storage_types.jl:1,  src/gp/lti_sde.jl:12-16
#   AbstractLGSSM interface                    src/models/lgssm.jl:1, test/test_util.jl:71-155
#   build_lgssm(f::LTISDE, x, Σys)             src/gp/lti_sde.jl:71-80
# Callers that stay UNCHANGED: src/gp/lti_sde.jl:33-68, src/gp/posterior_lti_sde.jl:18-78.
module TemporalGPsHIP

using TemporalGPs, AbstractGPs, FillArrays, StaticArrays, LinearAlgebra, Random
import TemporalGPs: StorageType, AbstractLGSSM, LTISDE, build_lgssm, lgssm_components, get_mean, get_kernel,
    posterior, marginals_diag, replace_observation_noise_cov, _filter, x0, ordering, Forward, Reverse, Gaussian,
    SArrayStorage, ArrayStorage

const libtgp = get(ENV, "TGP_HIP_LIB", "libtgp_hip.so")

# flags (include/tgp_hip.h)
const SHARED_A, SHARED_a, SHARED_Q, SHARED_H, SHARED_h, SHARED_R = (UInt32(1) << i for i in 0:5)

struct HIPStorage{T<:Real} <: StorageType{T}
    device::Int
end
HIPStorage(::Type{Float64}=Float64; device::Int=0) = HIPStorage{Float64}(device)

# one process per GPU: the calling thread (Julia's main thread) onto the CPUs next to the device, once per process and before the first handle
# (include/tgp_hip.h tgp_bind_host_thread; a thread on the far socket of a two-socket host pays ~13 us per logpdf + posterior pair)
bind_host_thread(device::Integer=0) = ccall((:tgp_bind_host_thread, libtgp), Cint, (Cint,), device) == 0

mutable struct Handle
    ptr::Ptr{Cvoid}
    function Handle(device::Integer)
        r = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:tgp_create, libtgp), Cint, (Ref{Ptr{Cvoid}}, Cint), r, device)
        rc == 0 || error("tgp_create failed with code $rc (is an MI355X visible?)")
        h = new(r[])
        finalizer(x -> ccall((:tgp_destroy, libtgp), Cint, (Ptr{Cvoid},), x.ptr), h)
        return h
    end
end

function check(h::Handle, rc::Cint)
    rc == 0 && return nothing
    msg = unsafe_string(ccall((:tgp_last_error, libtgp), Cstring, (Ptr{Cvoid},), h.ptr))
    rc == 2 && throw(PosDefException(0))          # mirrors cholesky / sqrt failures on the CPU path
    throw(error("libtgp_hip error $rc: $msg"))
end

"""Device-backed LGSSM: the flat column-major blocks live in `bufs` (host) and are uploaded once."""
struct DeviceLGSSM{Tord} <: AbstractLGSSM
    ordering::Tord
    h::Handle
    T::Int
    d::Int
    p::Int                # observations per time step (1: scalar outputs; > 1: SmallOutputLGC / space-time models)
    bufs::NamedTuple      # A, a, Q, H, hh, R :: Vector{Float64}; keeps the host copies alive
    flags::UInt32
    x0::Gaussian
    device::Int           # the GPU this model lives on: every model derived from it stays there
end

Base.length(m::DeviceLGSSM) = m.T
Base.eachindex(m::DeviceLGSSM{Forward}) = 1:m.T
Base.eachindex(m::DeviceLGSSM{Reverse}) = reverse(1:m.T)
TemporalGPs.ordering(m::DeviceLGSSM) = m.ordering
TemporalGPs.x0(m::DeviceLGSSM) = m.x0
TemporalGPs.storage_type(::DeviceLGSSM) = HIPStorage(Float64)

# Fill => one shared block + TGP_SHARED_* bit; Vector => per-step blocks (lti_sde.jl:135-160)
_flat(x::Fill) = (collect(Float64, vec(Array(first(x)))), true)
_flat(x::AbstractVector) = (reduce(vcat, (collect(Float64, vec(Array(xi))) for xi in x)), false)
_flat(x::Fill{<:Real}) = ([Float64(first(x))], true)
_flat(x::AbstractVector{<:Real}) = (collect(Float64, x), false)

# emission rows: the ABI wants H as [p][d] per step, row j = the j-th observation functional (row-major), i.e. the column-major
# layout of H'. Scalar outputs carry an adjoint d-vector (p = 1); vector outputs (SmallOutputLGC, space-time models) a p x d matrix.
_flat_rows(x::Fill) = (collect(Float64, vec(permutedims(Array(first(x))))), true)
_flat_rows(x::AbstractVector) = (reduce(vcat, (collect(Float64, vec(permutedims(Array(xi)))) for xi in x)), false)
# diagonal of the emission noise: a real per step (scalar outputs) or a Diagonal / vector of p variances (vector outputs)
_diag(x::Real) = Float64[x]
_diag(x::Diagonal) = collect(Float64, x.diag)
_diag(x::AbstractVector{<:Real}) = collect(Float64, x)
_flat_diag(x::Fill) = (_diag(first(x)), true)
_flat_diag(x::AbstractVector) = (reduce(vcat, (_diag(xi) for xi in x)), false)

function DeviceLGSSM(ord, As, as, Qs, Hs, hs, Σs, x0::Gaussian, device::Int)
    T, d = length(As), length(first(as))
    p = length(first(hs))          # 1 for ScalarOutputLGC; the number of observations per time step otherwise
    (A, sA), (a, sa), (Q, sQ) = _flat(As), _flat(as), _flat(Qs)
    (H, sH) = p == 1 ? _flat(Hs) : _flat_rows(Hs)
    (hh, sh) = _flat(hs)
    (R, sR) = p == 1 ? _flat(Σs) : _flat_diag(Σs)
    flags = UInt32(0)
    for (bit, s) in zip((SHARED_A, SHARED_a, SHARED_Q, SHARED_H, SHARED_h, SHARED_R), (sA, sa, sQ, sH, sh, sR))
        s && (flags |= bit)
    end
    h = Handle(device)
    x0m, x0P = collect(Float64, x0.m), collect(Float64, vec(Array(x0.P)))
    GC.@preserve A a Q H hh R x0m x0P begin
        check(h, ccall((:tgp_model_set, libtgp), Cint,
            (Ptr{Cvoid}, Int64, Cint, Cint, Cint, UInt32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
             Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
            h.ptr, T, d, p, ord isa Forward ? 0 : 1, flags, A, a, Q, H, hh, R, x0m, x0P))
    end
    return DeviceLGSSM(ord, h, T, d, p, (; A, a, Q, H, hh, R), flags, x0, device)
end

# The one method that selects the backend: components are built by the reference's own host code
# (SArrayStorage flavour), then packed (lti_sde.jl:71-80).
function TemporalGPs.build_lgssm(f::LTISDE{<:GP,<:HIPStorage}, x::AbstractVector, Σys::AbstractVector)
    As, as, Qs, (Hs, hs), x0 = lgssm_components(get_mean(f), get_kernel(f), x, SArrayStorage(Float64))
    return DeviceLGSSM(Forward(), As, as, Qs, Hs, hs, Σys, x0, f.storage.device)
end

# Space-time GPs (Separable kernels on a RectilinearGrid / RegularInTime): the reference's own dense components
# (space_time/to_gauss_markov.jl:1-20, ArrayStorage) go to the same constructor; d = Nr * d_t > 16 binds the dense fp64-MFMA engine
# behind the same entry points (tgp_model_set picks it by state dimension).
function TemporalGPs.build_lgssm(f::LTISDE{<:GP,<:HIPStorage}, x::TemporalGPs.SpaceTimeGrid, Σys::AbstractVector)
    As, as, Qs, (Hs, hs), x0 = lgssm_components(get_mean(f), get_kernel(f), x, ArrayStorage(Float64))
    return DeviceLGSSM(Forward(), As, as, Qs, Hs, hs, Σys, x0, f.storage.device)
end

# (values, what to hand to ccall for the `missing` argument, the array to GC.@preserve). The mask ARRAY is passed to ccall
# (converted to Ptr{UInt8} inside the preserved region), never a raw pointer taken outside it.
_split_missing(y::AbstractVector{<:Real}) = (collect(Float64, y), Ptr{UInt8}(C_NULL), nothing)
function _split_missing(y::AbstractVector{Union{Missing,T}}) where {T<:Real}
    mask = UInt8.(ismissing.(y))
    return (Float64[ismissing(v) ? 0.0 : v for v in y], mask, mask)
end

# vector observations: y[t] is a p-vector (possibly with missing entries, missings.jl:25-40); the ABI takes [T][p] = a p x T matrix
function _split_missing(y::AbstractVector{<:AbstractVector})
    Y = reduce(hcat, y)
    any(ismissing, Y) || return (collect(Float64, vec(Y)), Ptr{UInt8}(C_NULL), nothing)
    mask = UInt8.(vec(ismissing.(Y)))
    return (Float64[ismissing(v) ? 0.0 : v for v in vec(Y)], mask, mask)
end

# [T][p] results of the device as the reference's container: reals for scalar outputs, p-vectors otherwise
_per_step(v::Vector{Float64}, m) = m.p == 1 ? v : [v[(t-1)*m.p+1:t*m.p] for t in 1:m.T]

function AbstractGPs.logpdf(m::DeviceLGSSM, y::AbstractVector)
    length(m) == length(y) || throw(error("Dimension mismatch. length(prior) is $(length(m)), but length(y) is $(length(y))"))
    yv, mp, mask = _split_missing(y)
    out = Ref{Float64}(0.0)
    GC.@preserve yv mask check(m.h, ccall((:tgp_logpdf, libtgp), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{UInt8}, UInt32, Ref{Float64}), m.h.ptr, yv, mp, UInt32(0), out))
    return out[]
end

function TemporalGPs._filter(m::DeviceLGSSM, y::AbstractVector)
    yv, mp, mask = _split_missing(y)
    ms, Ps = Matrix{Float64}(undef, m.d, m.T), Array{Float64,3}(undef, m.d, m.d, m.T)
    GC.@preserve yv mask check(m.h, ccall((:tgp_filter, libtgp), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{UInt8}, UInt32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        m.h.ptr, yv, mp, UInt32(0), ms, Ps, C_NULL))
    return [Gaussian(ms[:, t], Ps[:, :, t]) for t in 1:m.T]
end

"""`posterior(model, ys)` on this backend is LAZY. The reference's unchanged callers (posterior_lti_sde.jl:27-36, :50-58, :62-78)
chain `replace_observation_noise_cov(posterior(model, ys), Σs_new)` into `marginals` / `rand` / `logpdf`;
`replace_observation_noise_cov` on a DevicePosterior only records Σs_new, and `marginals` of the result is ONE fused
filter + RTS smoother call (`tgp_posterior_marginals`): nothing of size T x (2 d^2 + d) crosses PCIe. `rand`, `logpdf`,
`_filter`, `getindex`-style access evaluate the reverse-time model once (`tgp_posterior`) on the SAME device."""
mutable struct DevicePosterior <: AbstractLGSSM
    prior::DeviceLGSSM{Forward}
    y::AbstractVector
    Σs_new::Union{Nothing,AbstractVector}
    model::Union{Nothing,DeviceLGSSM{Reverse}}
end

function TemporalGPs.posterior(m::DeviceLGSSM{Forward}, y::AbstractVector)
    length(m) == length(y) || throw(error("Dimension mismatch. length(prior) is $(length(m)), but length(y) is $(length(y))"))
    return DevicePosterior(m, y, nothing, nothing)
end

Base.length(p::DevicePosterior) = p.prior.T
Base.eachindex(p::DevicePosterior) = reverse(1:p.prior.T)
TemporalGPs.ordering(::DevicePosterior) = Reverse()
TemporalGPs.storage_type(::DevicePosterior) = HIPStorage(Float64)
TemporalGPs.x0(p::DevicePosterior) = materialise(p).x0

TemporalGPs.replace_observation_noise_cov(p::DevicePosterior, Σs_new::AbstractVector) =
    p.model === nothing ? DevicePosterior(p.prior, p.y, Σs_new, nothing) : replace_observation_noise_cov(p.model, Σs_new)

_prior_noise(m::DeviceLGSSM) = m.p == 1 ? (m.flags & SHARED_R != 0 ? Fill(m.bufs.R[1], m.T) : m.bufs.R) :
    (m.flags & SHARED_R != 0 ? Fill(m.bufs.R[1:m.p], m.T) : [m.bufs.R[(t-1)*m.p+1:t*m.p] for t in 1:m.T])

function AbstractGPs.marginals(p::DevicePosterior)
    p.model === nothing || return marginals(p.model)
    mean, var = posterior_marginals(p.prior, p.y, p.Σs_new === nothing ? _prior_noise(p.prior) : p.Σs_new)
    return _marginal_gaussians(p.prior, mean, var)
end
# rand of a posterior that has not been evaluated: `tgp_posterior_rand` (the filter and the reverse-time draw in one kernel, nothing of size
# T x (2 d^2 + d) written; Forward LTI models with scalar observations, d <= 6, no missing data) -- TGP_EUNSUPPORTED (4): the evaluated route
function AbstractGPs.rand(rng::AbstractRNG, p::DevicePosterior)
    m = p.prior
    if p.model === nothing && m.p == 1 && m.d <= 6 && !any(ismissing, p.y)
        # the randomness in the reference's order (lgssm.jl:65-77): T transition vectors, T emission scalars, then x0's
        eps_t = randn(rng, m.d, m.T); eps_e = randn(rng, m.T); eps_0 = randn(rng, m.d)
        Σ = p.Σs_new === nothing ? _prior_noise(m) : p.Σs_new
        Rn = Σ isa Fill ? Float64[first(Σ)] : collect(Float64, Σ)
        yv = collect(Float64, p.y)
        out = Vector{Float64}(undef, m.T)
        rc = GC.@preserve yv Rn ccall((:tgp_posterior_rand, libtgp), Cint,
            (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, UInt32, Ptr{Float64}),
            m.h.ptr, yv, Rn, eps_t, eps_e, eps_0, length(Rn) == 1 ? SHARED_R : UInt32(0), out)
        rc == 0 && return out
        rc == 4 || check(m.h, rc)
        # (not a model of the one-launch path: the same draws through the evaluated model)
        y = Vector{Float64}(undef, m.T)
        pm = materialise(p)
        check(pm.h, ccall((:tgp_rand, libtgp), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, UInt32, Ptr{Float64}),
            pm.h.ptr, eps_t, eps_e, eps_0, UInt32(0), y))
        return y
    end
    return rand(rng, materialise(p))
end
# logpdf of a posterior that has not been evaluated (posterior_lti_sde.jl:62-78's last line): log p(y* | y) = log p(y, y*) - log p(y), and two
# observations of one latent value with independent noise are one observation of it (DESIGN 3.18) -- two tgp_logpdf calls of the PRIOR, no
# reverse-time model of T x (2 d^2 + d) doubles evaluated or filtered.  (`lgssm.py::_posterior_logpdf_pair` is this function.)
function AbstractGPs.logpdf(p::DevicePosterior, y::AbstractVector)
    m = p.prior
    (p.model === nothing && m.p == 1 && length(y) == m.T) || return logpdf(materialise(p), y)
    R = _prior_noise(m)
    Rs = p.Σs_new === nothing ? R : p.Σs_new
    ȳ = Vector{Union{Missing,Float64}}(missing, m.T)
    R̄ = Vector{Float64}(undef, m.T)
    pair = 0.0
    for t in 1:m.T
        a, b, r, rs = p.y[t], y[t], Float64(R[t]), Float64(Rs[t])
        if !ismissing(a) && !ismissing(b)
            pair -= (log(2π * (r + rs)) + (a - b)^2 / (r + rs)) / 2
            ȳ[t], R̄[t] = (rs * a + r * b) / (r + rs), r * rs / (r + rs)
        elseif !ismissing(a)
            ȳ[t], R̄[t] = a, r
        else
            R̄[t] = rs                    # (missing on both sides stays missing)
            ismissing(b) || (ȳ[t] = b)
        end
    end
    Σ̄ = all(==(R̄[1]), R̄) ? Fill(R̄[1], m.T) : R̄     # one variance keeps the model on its one-launch path
    ȳv = any(ismissing, ȳ) ? ȳ : Float64.(ȳ)
    if Σ̄ isa Fill && ȳv isa Vector{Float64}             # the prior with another noise variance, on the prior's own handle (tgp_logpdf_noise)
        out = Ref{Float64}(0.0)
        rc = GC.@preserve ȳv ccall((:tgp_logpdf_noise, libtgp), Cint, (Ptr{Cvoid}, Ptr{Float64}, UInt32, Float64, Ref{Float64}),
                                   m.h.ptr, ȳv, UInt32(0), R̄[1], out)
        rc == 0 && return out[] + pair - logpdf(m, p.y)
        rc == 4 || check(m.h, rc)                        # TGP_EUNSUPPORTED: bind the joint model
    end
    return logpdf(replace_observation_noise_cov(m, Σ̄), ȳv) + pair - logpdf(m, p.y)
end
TemporalGPs._filter(p::DevicePosterior, y::AbstractVector) = _filter(materialise(p), y)

"""Evaluate the reverse-time model (lgssm.jl:193-238) once, on the prior's device."""
function materialise(p::DevicePosterior)
    p.model === nothing || return p.model
    m = p.prior
    haskey(m.bufs, :A) || error("posterior of an SDE-described model: only marginals(replace_observation_noise_cov(posterior(m, y), Σ)) is available (the per-step A_k, Q_k exist on the device only)")
    yv, mp, mask = _split_missing(p.y)
    G, g, L = Array{Float64,3}(undef, m.d, m.d, m.T), Matrix{Float64}(undef, m.d, m.T), Array{Float64,3}(undef, m.d, m.d, m.T)
    xfm, xfP = Vector{Float64}(undef, m.d), Matrix{Float64}(undef, m.d, m.d)
    GC.@preserve yv mask check(m.h, ccall((:tgp_posterior, libtgp), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{UInt8}, UInt32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        m.h.ptr, yv, mp, UInt32(0), G, g, L, xfm, xfP))
    Gs, gs, Ls = [G[:, :, t] for t in 1:m.T], [g[:, t] for t in 1:m.T], [L[:, :, t] for t in 1:m.T]
    Hs, hs = _emission_blocks(m)
    Rs = p.Σs_new === nothing ? _prior_noise(m) : p.Σs_new
    p.model = DeviceLGSSM(Reverse(), Gs, gs, Ls, Hs, hs, Rs, Gaussian(xfm, xfP), m.device)
    return p.model
end

# the packed emission blocks back as per-step containers the constructor accepts (H rows are stored row-major: [p][d])
function _emission_blocks(m::DeviceLGSSM)
    d, p, T = m.d, m.p, m.T
    Hblk(v) = p == 1 ? v[1:d] : permutedims(reshape(v[1:p*d], d, p))
    Hs = m.flags & SHARED_H != 0 ? Fill(Hblk(m.bufs.H), T) : [Hblk(@view m.bufs.H[(t-1)*p*d+1:t*p*d]) for t in 1:T]
    hblk(v) = p == 1 ? v[1] : v[1:p]
    hs = m.flags & SHARED_h != 0 ? Fill(hblk(m.bufs.hh), T) : [hblk(@view m.bufs.hh[(t-1)*p+1:t*p]) for t in 1:T]
    return Hs, hs
end

function TemporalGPs.replace_observation_noise_cov(m::DeviceLGSSM, Σs_new::AbstractVector)
    # re-bind the model with the new noise ON THE SAME DEVICE; the transition blocks are reused as they are
    d, T = m.d, m.T
    if haskey(m.bufs, :F)      # SDE-described transitions: re-describe them, the device rebuilds A_k, Q_k
        return DeviceLGSSM_sde(reshape(m.bufs.F, d, d), m.bufs.a, m.bufs.H, m.bufs.hh[1], Σs_new, m.bufs.times,
                               reshape(m.bufs.A1, d, d), reshape(m.bufs.Q1, d, d), m.x0, m.device)
    end
    blk(v, n, shared) = shared ? Fill(v[1:n], T) : [v[(t-1)*n+1:t*n] for t in 1:T]
    mat(v, shared) = shared ? Fill(reshape(v[1:d*d], d, d), T) : [reshape(v[(t-1)*d*d+1:t*d*d], d, d) for t in 1:T]
    Hs, hs = _emission_blocks(m)
    return DeviceLGSSM(m.ordering, mat(m.bufs.A, m.flags & SHARED_A != 0), blk(m.bufs.a, d, m.flags & SHARED_a != 0),
        mat(m.bufs.Q, m.flags & SHARED_Q != 0), Hs, hs, Σs_new, m.x0, m.device)
end

# marginals_diag semantics (lgssm.jl:128-137, linear_gaussian_conditionals.jl `posterior_and_lml` callers): mean and the DIAGONAL of
# the covariance of every emission; p-variate steps come back as Gaussians with Diagonal covariance
_marginal_gaussians(m, mean, var) = m.p == 1 ? [Gaussian(mean[t], var[t]) for t in 1:m.T] :
    [Gaussian(mean[(t-1)*m.p+1:t*m.p], Diagonal(var[(t-1)*m.p+1:t*m.p])) for t in 1:m.T]

function AbstractGPs.marginals(m::DeviceLGSSM)
    mean, var = Vector{Float64}(undef, m.p * m.T), Vector{Float64}(undef, m.p * m.T)
    check(m.h, ccall((:tgp_marginals, libtgp), Cint, (Ptr{Cvoid}, UInt32, Ptr{Float64}, Ptr{Float64}),
        m.h.ptr, UInt32(0), mean, var))
    return _marginal_gaussians(m, mean, var)      # marginals(::Gaussian) -> Normal(mean, sqrt(var)), gaussian.jl:61-63
end

function AbstractGPs.rand(rng::AbstractRNG, m::DeviceLGSSM)
    # randomness drawn in the reference's order (lgssm.jl:65-77): T transition vectors, T emission scalars, then x0
    eps_t = randn(rng, m.d, m.T); eps_e = randn(rng, m.p, m.T); eps_0 = randn(rng, m.d)
    y = Vector{Float64}(undef, m.p * m.T)
    check(m.h, ccall((:tgp_rand, libtgp), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, UInt32, Ptr{Float64}),
        m.h.ptr, eps_t, eps_e, eps_0, UInt32(0), y))
    return _per_step(y, m)
end

"""Fused `marginals(replace_observation_noise_cov(posterior(model, y), Σs_new))` (posterior_lti_sde.jl:27-36): nothing is
materialised on the host. The reference's unchanged caller reaches it through DevicePosterior (above)."""
function posterior_marginals(m::DeviceLGSSM{Forward}, y::AbstractVector, Σs_new::AbstractVector)
    yv, mp, mask = _split_missing(y)
    (R, shared) = m.p == 1 ? _flat(Σs_new) : _flat_diag(Σs_new)
    fl = shared ? SHARED_R : UInt32(0)
    mean, var = Vector{Float64}(undef, m.p * m.T), Vector{Float64}(undef, m.p * m.T)
    GC.@preserve yv mask R check(m.h, ccall((:tgp_posterior_marginals, libtgp), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{UInt8}, Ptr{Float64}, UInt32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        m.h.ptr, yv, mp, R, fl, mean, var, C_NULL))
    return mean, var
end

"""`logpdf` and its derivative along `P` tangent directions of the packed model blocks (tgp_logpdf_grad; Forward models
whose blocks are all shared -- RegularSpacing, homoscedastic noise). `tangents.dA` is `d*d x P` etc.; a Mooncake /
ChainRules rule for `logpdf(::DeviceLGSSM, y)` contracts the returned vector with the parameter -> block Jacobian."""
function logpdf_and_directional_derivatives(m::DeviceLGSSM{Forward}, y::AbstractVector, tangents::NamedTuple)
    yv, mp, mask = _split_missing(y)
    P = size(tangents.dA, 2)
    lml, grad = Ref{Float64}(0.0), Vector{Float64}(undef, P)
    t = map(x -> collect(Float64, vec(x)), tangents)
    GC.@preserve yv mask t check(m.h, ccall((:tgp_logpdf_grad, libtgp), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{UInt8}, UInt32, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
         Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Float64}, Ptr{Float64}),
        m.h.ptr, yv, mp, UInt32(0), P, t.dA, t.da, t.dQ, t.dH, t.dh, t.dR, t.dx0m, t.dx0P, lml, grad))
    return lml[], grad
end

"""`logpdf` and its gradient with respect to the packed model blocks by ONE adjoint pass (tgp_logpdf_adjoint: Forward models whose
blocks are all shared, one noise variance, scalar outputs, no missing data, d <= 8) -- the pullback a ChainRules / Mooncake rule for
`logpdf(::DeviceLGSSM, y)` needs: the rule returns `lml` and contracts the named tuple with the cotangent (the reference differentiates
the sequential loop itself, bench/single_output_gps.jl:149-156). Matrices come back d x d (column-major, as Julia stores them)."""
function logpdf_and_block_gradients(m::DeviceLGSSM{Forward}, y::AbstractVector{<:Real})
    d = m.d
    yv = collect(Float64, y)
    lml = Ref{Float64}(0.0)
    gA, gQ, gP = (Matrix{Float64}(undef, d, d) for _ in 1:3)
    ga, gH, gm = (Vector{Float64}(undef, d) for _ in 1:3)
    gh, gR = Ref{Float64}(0.0), Ref{Float64}(0.0)
    GC.@preserve yv check(m.h, ccall((:tgp_logpdf_adjoint, libtgp), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, UInt32, Ref{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Float64},
         Ref{Float64}, Ptr{Float64}, Ptr{Float64}),
        m.h.ptr, yv, UInt32(0), lml, gA, ga, gQ, gH, gh, gR, gm, gP))
    return lml[], (A = gA, a = ga, Q = gQ, H = gH, h = gh[], R = gR[], x0m = gm, x0P = gP)
end

# ---- one series over the GPUs of a node, one process (tgp_create_multi: one handle + stream + RCCL communicator per device inside
#      the library; SURVEY.md 8b "Threading", 8e). Rank r owns the contiguous segment tgp_multi_segment gives it; inputs and outputs
#      are addressed per segment (pointer + offset into the caller's arrays: nothing is copied on the host).
mutable struct MultiHandle
    ptr::Ptr{Cvoid}
    ndev::Int
    function MultiHandle(devices::AbstractVector{<:Integer})
        r = Ref{Ptr{Cvoid}}(C_NULL)
        devs = collect(Cint, devices)
        rc = ccall((:tgp_create_multi, libtgp), Cint, (Ref{Ptr{Cvoid}}, Cint, Ptr{Cint}), r, length(devs), devs)
        rc == 0 || throw(error("tgp_create_multi failed ($rc)"))
        h = new(r[], length(devs))
        finalizer(x -> ccall((:tgp_destroy_multi, libtgp), Cint, (Ptr{Cvoid},), x.ptr), h)
        return h
    end
end

function check(h::MultiHandle, rc::Cint)
    rc == 0 && return nothing
    msg = unsafe_string(ccall((:tgp_multi_last_error, libtgp), Cstring, (Ptr{Cvoid},), h.ptr))
    rc == 1 && occursin("Dimension mismatch", msg) && throw(DimensionMismatch(msg))
    throw(error("libtgp_hip error $rc: $msg"))
end

"""A DeviceLGSSM whose time axis is sharded over `devices` (Forward, diagonal noise, d <= 16)."""
struct MultiDeviceLGSSM <: AbstractLGSSM
    h::MultiHandle
    T::Int
    d::Int
    p::Int
    bufs::NamedTuple
    x0::Gaussian
end
Base.length(m::MultiDeviceLGSSM) = m.T
Base.eachindex(m::MultiDeviceLGSSM) = 1:m.T
TemporalGPs.ordering(::MultiDeviceLGSSM) = Forward()
TemporalGPs.x0(m::MultiDeviceLGSSM) = m.x0
TemporalGPs.storage_type(::MultiDeviceLGSSM) = HIPStorage(Float64)

function MultiDeviceLGSSM(As, as, Qs, Hs, hs, Σs, x0::Gaussian, devices::AbstractVector{<:Integer})
    T, d = length(As), length(first(as))
    p = length(first(hs))
    (A, sA), (a, sa), (Q, sQ) = _flat(As), _flat(as), _flat(Qs)
    (H, sH) = p == 1 ? _flat(Hs) : _flat_rows(Hs)
    (hh, sh) = _flat(hs)
    (R, sR) = p == 1 ? _flat(Σs) : _flat_diag(Σs)
    flags = UInt32(0)
    for (bit, s) in zip((SHARED_A, SHARED_a, SHARED_Q, SHARED_H, SHARED_h, SHARED_R), (sA, sa, sQ, sH, sh, sR))
        s && (flags |= bit)
    end
    h = MultiHandle(devices)
    x0m, x0P = collect(Float64, x0.m), collect(Float64, vec(Array(x0.P)))
    GC.@preserve A a Q H hh R x0m x0P check(h, ccall((:tgp_multi_model_set, libtgp), Cint,
        (Ptr{Cvoid}, Int64, Cint, Cint, Cint, UInt32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
         Ptr{Float64}, Ptr{Float64}), h.ptr, T, d, p, 0, flags, A, a, Q, H, hh, R, x0m, x0P))
    return MultiDeviceLGSSM(h, T, d, p, (; A, a, Q, H, hh, R), x0)
end

"""`DeviceLGSSM(...; ndev)`: the same components, sharded over the first `ndev` GPUs."""
DeviceLGSSM(ord::Forward, As, as, Qs, Hs, hs, Σs, x0::Gaussian; ndev::Int) = MultiDeviceLGSSM(As, as, Qs, Hs, hs, Σs, x0, 0:(ndev - 1))

# per-rank addresses of one host array: element offset t0_r * width
function _parts(m::MultiDeviceLGSSM, v::Vector{Float64}, width::Int)
    t0, t1 = Ref{Int64}(0), Ref{Int64}(0)
    return [begin
                ccall((:tgp_multi_segment, libtgp), Cint, (Int64, Cint, Cint, Ref{Int64}, Ref{Int64}), m.T, m.h.ndev, r, t0, t1)
                pointer(v, t0[] * width + 1)
            end for r in 0:(m.h.ndev - 1)]
end

function AbstractGPs.logpdf(m::MultiDeviceLGSSM, y::AbstractVector{<:Real})
    yv, out = collect(Float64, y), Ref{Float64}(0.0)
    GC.@preserve yv begin
        ys = _parts(m, yv, m.p)
        check(m.h, ccall((:tgp_multi_logpdf, libtgp), Cint, (Ptr{Cvoid}, Ptr{Ptr{Float64}}, Ptr{Ptr{UInt8}}, UInt32, Ref{Float64}),
            m.h.ptr, ys, C_NULL, UInt32(0), out))
    end
    return out[]
end

"""(logpdf, mean, var) of `marginals(replace_observation_noise_cov(posterior(model, y), Σs_new))` over all the GPUs of the handle."""
function logpdf_and_posterior_marginals(m::MultiDeviceLGSSM, y::AbstractVector{<:Real}, Σs_new::AbstractVector)
    yv = collect(Float64, y)
    (R, shared) = m.p == 1 ? _flat(Σs_new) : _flat_diag(Σs_new)
    mean, var = Vector{Float64}(undef, m.p * m.T), Vector{Float64}(undef, m.p * m.T)
    lml = Ref{Float64}(0.0)
    GC.@preserve yv R mean var begin
        ys, ms, vs = _parts(m, yv, m.p), _parts(m, mean, m.p), _parts(m, var, m.p)
        Rs = shared ? fill(pointer(R), m.h.ndev) : _parts(m, R, m.p)
        check(m.h, ccall((:tgp_multi_logpdf_and_posterior_marginals, libtgp), Cint,
            (Ptr{Cvoid}, Ptr{Ptr{Float64}}, Ptr{Ptr{UInt8}}, Ptr{Ptr{Float64}}, UInt32, Ref{Float64}, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}),
            m.h.ptr, ys, C_NULL, Rs, shared ? SHARED_R : UInt32(0), lml, ms, vs))
    end
    return lml[], mean, var
end
posterior_marginals(m::MultiDeviceLGSSM, y::AbstractVector{<:Real}, Σs_new::AbstractVector) = logpdf_and_posterior_marginals(m, y, Σs_new)[2:3]

"""Irregularly spaced inputs without host-side matrix exponentials (tgp_model_set_sde): the device evaluates
A_k = exp(F dt_k), Q_k = P_inf - A_k P_inf A_k' (lti_sde.jl:135-146) from F, P_inf and the time stamps.
`A1`, `Q1` override the first transition (the reference fixes dt_1 := 1 per kernel component, lti_sde.jl:139)."""
function DeviceLGSSM_sde(F, a, H, hh, Σs, times::AbstractVector{<:Real}, A1, Q1, x0::Gaussian, device::Int)
    T, d = length(times), length(a)
    (R, sR) = _flat(Σs)
    flags = SHARED_a | SHARED_H | SHARED_h | (sR ? SHARED_R : UInt32(0))
    h = Handle(device)
    Fv, av, Hv, hv = collect(Float64, vec(F)), collect(Float64, a), collect(Float64, H), [Float64(hh)]
    tv, A1v, Q1v = collect(Float64, times), collect(Float64, vec(A1)), collect(Float64, vec(Q1))
    x0m, x0P = collect(Float64, x0.m), collect(Float64, vec(Array(x0.P)))
    GC.@preserve Fv av Hv hv R tv A1v Q1v x0m x0P begin
        check(h, ccall((:tgp_model_set_sde, libtgp), Cint,
            (Ptr{Cvoid}, Int64, Cint, Cint, UInt32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
             Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
            h.ptr, T, d, 0, flags, Fv, av, Hv, hv, R, tv, A1v, Q1v, x0m, x0P))
    end
    return DeviceLGSSM(Forward(), h, T, d, 1, (; F = Fv, a = av, H = Hv, hh = hv, R, times = tv, A1 = A1v, Q1 = Q1v), flags, x0, device)
end

"""Posterior marginals through other emissions (tgp_posterior_marginals_at): the fast path of
`approx_posterior_marginals` (space_time/pseudo_point.jl:198-235). `Hn` is `pn x d`, `Σs_new` a vector of length-`pn` diagonals."""
function posterior_marginals_at(m::DeviceLGSSM{Forward}, y::AbstractVector, Hn::AbstractMatrix, hn::AbstractVector, Σs_new::AbstractVector)
    yv, mp, mask = _split_missing(y)
    pn = size(Hn, 1)
    H = collect(Float64, vec(permutedims(Hn)))            # row-major [pn][d]
    hv = collect(Float64, hn)
    R, fl = Σs_new isa Fill ? (collect(Float64, first(Σs_new)), SHARED_R) : (reduce(vcat, (collect(Float64, s) for s in Σs_new)), UInt32(0))
    mean, var = Matrix{Float64}(undef, pn, m.T), Matrix{Float64}(undef, pn, m.T)
    GC.@preserve yv mask H hv R check(m.h, ccall((:tgp_posterior_marginals_at, libtgp), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{UInt8}, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, UInt32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        m.h.ptr, yv, mp, pn, H, hv, R, fl, mean, var, C_NULL))
    return mean, var                                       # column t = the pn marginals of time step t
end

end # module
