module BOSSHip
using BOSS, LinearAlgebra, ForwardDiff
import BOSS: model_posterior_slice, data_loglike, params_loglike, _params_sampler, vectorizer, bijector,
             sliceable, slice, join_slices, make_discrete, mean, var, mean_and_var,
             estimate_parameters, maximize_acquisition

const lib = joinpath(@__DIR__, "libbosship.so")
check(rc) = rc == 0 ? nothing :
    rc == 3 ? throw(PosDefException(0)) :
    rc == 4 ? throw(DomainError(NaN, unsafe_string(ccall((:boss_last_error, lib), Cstring, ())))) :
              error(unsafe_string(ccall((:boss_last_error, lib), Cstring, ())))
kernel_id(k) = k isa BOSS.Matern32Kernel ? 0 : k isa BOSS.Matern52Kernel ? 1 : 2      # SqExponential / Gaussian
version() = unsafe_string(ccall((:boss_version, lib), Cstring, ()))
device_count() = (n = Ref{Cint}(); check(ccall((:boss_device_count, lib), Cint, (Ref{Cint},), n)); Int(n[]))
"All GPUs of the node from this process: contexts + one RCCL communicator per device.  Idempotent; returns (devices, exchanges over RCCL)."
function init_devices()
    n = Ref{Cint}(); check(ccall((:boss_init, lib), Cint, (Ref{Cint},), n))
    nd = Ref{Cint}(); rccl = Ref{Cint}(); check(ccall((:boss_comm_info, lib), Cint, (Ref{Cint}, Ref{Cint}), nd, rccl))
    return Int(nd[]), rccl[] != 0
end
function __init__()
    atexit(() -> ccall((:boss_shutdown, lib), Cvoid, ()))            # communicators and exchange buffers (handles go with their finalizers)
end

# ---------------------------------------------------------------- SurrogateModel
"GaussianProcess whose posterior lives on an MI355X.  Wraps a BOSS.GaussianProcess for its mean, kernel and priors."
struct HipGaussianProcess{G<:BOSS.GaussianProcess} <: BOSS.SurrogateModel
    gp::G
    device::Cint
end
HipGaussianProcess(gp::BOSS.GaussianProcess; device = 0) = HipGaussianProcess(gp, Cint(device))

"""
The parameters of `HipGaussianProcess`.  It MUST be its own type: `update_parameters!` asserts
`typeof(problem.model) <: M` for `FittedParams{M}` (src/types/problem.jl:177-179), and `MAPParams{M}` takes its `M` from
`ModelParams{M}` (src/types/parameters.jl:118-123) — an alias of `GaussianProcessParams <: ModelParams{GaussianProcess}`
would fail that assert under an unmodified `bo!`.
"""
struct HipGPParams{L<:AbstractMatrix{<:Real}, A<:AbstractVector{<:Real}, N<:AbstractVector{<:Real}} <: BOSS.ModelParams{HipGaussianProcess}
    λ::L
    α::A
    σ::N
end
to_gp(p::HipGPParams) = BOSS.GaussianProcessParams(p.λ, p.α, p.σ)
to_hip(p::BOSS.GaussianProcessParams) = HipGPParams(p.λ, p.α, p.σ)

# the model API of src/types/surrogate_model.jl:19-73, delegated to the wrapped GaussianProcess with the parameter
# container converted at the boundary (gaussian_process.jl:85-119, :282-343)
params_loglike(m::HipGaussianProcess) = (ll = params_loglike(m.gp); (p::HipGPParams) -> ll(to_gp(p)))
_params_sampler(m::HipGaussianProcess) = (sample = _params_sampler(m.gp); rng -> to_hip(sample(rng)))
function vectorizer(m::HipGaussianProcess)
    vec_gp, devec_gp = vectorizer(m.gp)                    # skips Dirac-prior parameters (src/models/utils/dirac.jl:36-77)
    vectorize(p::HipGPParams) = vec_gp(to_gp(p))
    devectorize(p::HipGPParams, ps::AbstractVector{<:Real}) = to_hip(devec_gp(to_gp(p), ps))
    return vectorize, devectorize
end
bijector(m::HipGaussianProcess) = bijector(m.gp)           # acts on the vectorised parameters: no container involved
BOSS.param_priors(m::HipGaussianProcess) = BOSS.param_priors(m.gp)
BOSS.param_count(p::HipGPParams) = sum(BOSS.param_lengths(p))
BOSS.param_lengths(p::HipGPParams) = (length(p.λ), length(p.α), length(p.σ))
BOSS.param_shapes(p::HipGPParams) = (size(p.λ), size(p.α), size(p.σ))
sliceable(::HipGaussianProcess) = true
slice(m::HipGaussianProcess, i::Int) = HipGaussianProcess(slice(m.gp, i), m.device)
slice(p::HipGPParams, i::Int) = HipGPParams(p.λ[:, i:i], p.α[i:i], p.σ[i:i])
join_slices(ps::AbstractVector{<:HipGPParams}) =
    HipGPParams(hcat(getfield.(ps, Ref(:λ))...), vcat(getfield.(ps, Ref(:α))...), vcat(getfield.(ps, Ref(:σ))...))
make_discrete(m::HipGaussianProcess, d::AbstractVector{Bool}) = HipGaussianProcess(make_discrete(m.gp, d), m.device)

"A device handle owned by Julia: freed by the garbage collector (or `close`)."
mutable struct Handle
    h::Ptr{Cvoid}
    function Handle(h::Ptr{Cvoid})
        obj = new(h)
        finalizer(close, obj)
    end
end
Base.close(o::Handle) = (o.h == C_NULL || ccall((:boss_gp_free, lib), Cvoid, (Ptr{Cvoid},), o.h); o.h = C_NULL; nothing)
Base.unsafe_convert(::Type{Ptr{Cvoid}}, o::Handle) = o.h

struct HipPosteriorSlice <: BOSS.ModelPosteriorSlice{HipGaussianProcess}
    h::Handle
    mean                                                           # prior mean of this output: nothing, a Real or x -> Real
end
mean_vals(::Nothing, X) = C_NULL
mean_vals(m::Real, X) = fill(Float64(m), size(X, 2))
mean_vals(m::Function, X) = Float64[m(x) for x in eachcol(X)]
has_mean(post::HipPosteriorSlice) = !isnothing(post.mean)
discrete_flags(k) = k isa BOSS.DiscreteKernel && !(k.dims isa Missing) ? UInt8.(k.dims) : C_NULL
base_kernel(k) = k isa BOSS.DiscreteKernel ? k.kernel : k

function model_posterior_slice(m::HipGaussianProcess, p::HipGPParams, data::BOSS.ExperimentData, i::Int)
    X = Matrix{Float64}(data.X); y = Vector{Float64}(data.Y[i, :])
    mu = BOSS.mean_getindex(m.gp.mean, i)                          # gaussian_process.jl:101-103
    h = Ref{Ptr{Cvoid}}(); lp = Ref{Cdouble}()
    check(ccall((:boss_gp_fit, lib), Cint,
        (Cint, Cint, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Cdouble,
         Ptr{UInt8}, Ref{Ptr{Cvoid}}, Ref{Cdouble}),
        m.device, kernel_id(base_kernel(m.gp.kernel)), size(X, 1), size(X, 2), X, y, mean_vals(mu, X),
        Vector{Float64}(p.λ[:, i]), p.α[i], p.σ[i], discrete_flags(m.gp.kernel), h, lp))
    return HipPosteriorSlice(Handle(h[]), mu)
end

function mean_and_var(post::HipPosteriorSlice, X::AbstractMatrix{<:Real})
    Xs = Matrix{Float64}(X); M = size(Xs, 2)
    μ = Vector{Float64}(undef, M); σ2 = similar(μ); bad = Ref{Clong}(-1)
    GC.@preserve post check(ccall((:boss_gp_predict, lib), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Clong}),
        post.h.h, M, Xs, mean_vals(post.mean, Xs), μ, σ2, bad))
    return μ, σ2
end
mean_and_var(post::HipPosteriorSlice, x::AbstractVector{<:Real}) = first.(mean_and_var(post, hcat(x)))
mean(post::HipPosteriorSlice, x) = mean_and_var(post, x)[1]
var(post::HipPosteriorSlice, x) = mean_and_var(post, x)[2]
function BOSS.mean_and_cov(post::HipPosteriorSlice, X::AbstractMatrix{<:Real})
    Xs = Matrix{Float64}(X); M = size(Xs, 2)
    μ = Vector{Float64}(undef, M); Σ = Matrix{Float64}(undef, M, M); bad = Ref{Clong}(-1)
    GC.@preserve post check(ccall((:boss_gp_predict_cov, lib), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Clong}),
        post.h.h, M, Xs, mean_vals(post.mean, Xs), μ, Σ, bad))
    return μ, Σ
end
BOSS.cov(post::HipPosteriorSlice, X::AbstractMatrix{<:Real}) = BOSS.mean_and_cov(post, X)[2]

function data_loglike(m::HipGaussianProcess, data::BOSS.ExperimentData)
    X = Matrix{Float64}(data.X)
    hs = map(1:size(data.Y, 1)) do i                                   # one resident handle per output, freed with the closure
        h = Ref{Ptr{Cvoid}}()
        check(ccall((:boss_gp_create, lib), Cint, (Cint, Cint, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{UInt8}, Ref{Ptr{Cvoid}}),
              m.device, kernel_id(base_kernel(m.gp.kernel)), size(X, 1), size(X, 2), X, Vector{Float64}(data.Y[i, :]),
              discrete_flags(m.gp.kernel), h))
        Handle(h[])
    end
    means = [mean_vals(BOSS.mean_getindex(m.gp.mean, i), X) for i in eachindex(hs)]   # the GP's prior mean has no parameters
    return function ll_data(p::HipGPParams)                            # the closure keeps `hs` alive; the finalizers free them
        for i in eachindex(hs)                                         # all outputs are enqueued (flag 1 = BOSS_FIT_NO_SYNC) ...
            check(ccall((:boss_gp_update, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cdouble, Cdouble, Ptr{Cdouble}, Cint, Ptr{Cdouble}),
                  hs[i].h, Vector{Float64}(p.λ[:, i]), p.α[i], p.σ[i], means[i], 1, C_NULL))
        end
        sum(eachindex(hs)) do i                                        # ... then collected: the device never waits for the host in between
            lp = Ref{Cdouble}()
            check(ccall((:boss_gp_sync, lib), Cint, (Ptr{Cvoid}, Ref{Cdouble}), hs[i].h, lp))
            lp[]
        end
    end                                    # exceptions → -Inf via BOSS.safe_data_loglike, as for any model
end

# ---------------------------------------------------------------- Semiparametric (parametric mean + GP; BASELINE config 4)
"BOSS.Semiparametric whose GP part lives on an MI355X: the parametric model is evaluated on the host into the mean vectors."
struct HipSemiparametric{S<:BOSS.Semiparametric} <: BOSS.SurrogateModel
    sp::S
    device::Cint
end
HipSemiparametric(sp::BOSS.Semiparametric; device = 0) = HipSemiparametric(sp, Cint(device))

"Own parameter type (same reason as `HipGPParams`): θ of the parametric mean, then the GP's λ, α, σ (semiparametric.jl:52-63)."
struct HipSemiparametricParams{T<:AbstractVector{<:Real}, L<:AbstractMatrix{<:Real}, A<:AbstractVector{<:Real},
                               N<:AbstractVector{<:Real}} <: BOSS.ModelParams{HipSemiparametric}
    θ::T
    λ::L
    α::A
    σ::N
end
to_sp(p::HipSemiparametricParams) = BOSS.SemiparametricParams(p.θ, p.λ, p.α, p.σ)
to_hip(p::BOSS.SemiparametricParams) = HipSemiparametricParams(p.θ, p.λ, p.α, p.σ)
gp_part(p::HipSemiparametricParams) = HipGPParams(p.λ, p.α, p.σ)                    # _extract_gp_params, semiparametric.jl:65-71

# model API (src/types/surrogate_model.jl:19-73) delegated to the wrapped model, containers converted at the boundary
params_loglike(m::HipSemiparametric) = (ll = params_loglike(m.sp); (p::HipSemiparametricParams) -> ll(to_sp(p)))
_params_sampler(m::HipSemiparametric) = (sample = _params_sampler(m.sp); rng -> to_hip(sample(rng)))
function vectorizer(m::HipSemiparametric)
    vec_sp, devec_sp = vectorizer(m.sp)                    # θ, vec(λ), α, σ with the Dirac-prior entries left out (semiparametric.jl:114-144)
    vectorize(p::HipSemiparametricParams) = vec_sp(to_sp(p))
    devectorize(p::HipSemiparametricParams, ps::AbstractVector{<:Real}) = to_hip(devec_sp(to_sp(p), ps))
    return vectorize, devectorize
end
bijector(m::HipSemiparametric) = bijector(m.sp)
BOSS.param_priors(m::HipSemiparametric) = BOSS.param_priors(m.sp)
BOSS.param_count(p::HipSemiparametricParams) = sum(BOSS.param_lengths(p))
BOSS.param_lengths(p::HipSemiparametricParams) = (length(p.θ), length(p.λ), length(p.α), length(p.σ))
BOSS.param_shapes(p::HipSemiparametricParams) = (size(p.θ), size(p.λ), size(p.α), size(p.σ))
sliceable(::HipSemiparametric) = false                    # θ is shared by all outputs (as for BOSS.Semiparametric)
make_discrete(m::HipSemiparametric, d::AbstractVector{Bool}) = HipSemiparametric(make_discrete(m.sp, d), m.device)

"The GP of the semiparametric model under θ: the nonparametric part with the parametric prediction as its mean (semiparametric.jl:79-85)."
hip_gp(m::HipSemiparametric, θ) = HipGaussianProcess(BOSS.add_mean(m.sp.nonparametric, m.sp.parametric(θ)), m.device)
model_posterior_slice(m::HipSemiparametric, p::HipSemiparametricParams, data::BOSS.ExperimentData, i::Int) =
    model_posterior_slice(hip_gp(m, p.θ), gp_part(p), data, i)   # a HipPosteriorSlice whose `mean` is x -> f(x; θ)[i]: mean_Xs follows from it

function data_loglike(m::HipSemiparametric, data::BOSS.ExperimentData)
    X = Matrix{Float64}(data.X); k = m.sp.nonparametric.kernel
    hs = map(1:size(data.Y, 1)) do i                                   # resident data, one handle per output
        h = Ref{Ptr{Cvoid}}()
        check(ccall((:boss_gp_create, lib), Cint, (Cint, Cint, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{UInt8}, Ref{Ptr{Cvoid}}),
              m.device, kernel_id(base_kernel(k)), size(X, 1), size(X, 2), X, Vector{Float64}(data.Y[i, :]), discrete_flags(k), h))
        Handle(h[])
    end
    return function ll_data(p::HipSemiparametricParams)
        f = m.sp.parametric(p.θ)                                       # semiparametric.jl:87-92: the mean changes with θ
        F = reduce(hcat, (Float64.(f(x)) for x in eachcol(X)))         # y_dim × N, evaluated once per likelihood call
        sum(eachindex(hs)) do i
            lp = Ref{Cdouble}()
            check(ccall((:boss_gp_update, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cdouble, Cdouble, Ptr{Cdouble}, Cint, Ref{Cdouble}),
                  hs[i].h, Vector{Float64}(p.λ[:, i]), p.α[i], p.σ[i], Vector{Float64}(F[i, :]), 0, lp))
            lp[]
        end
    end
end

# ---------------------------------------------------------------- Bayesian inference (BASELINE config 5)
# `BIParams` (src/types/parameters.jl:128-146) carries S parameter samples; `model_posterior(model, ::AbstractVector{<:ModelParams},
# data)` (src/posterior.jl:15-19) broadcasts over them, so `BOSS.model_posterior(problem)` of a problem fitted with TuringBI
# (ext/TuringExt.jl:88-107 — it only needs the model API above) is a Vector of posteriors = P×S device handles, and
# `HipBatchAM` averages the acquisition over the S samples on the device (`boss_acq_ei(P, S, …)`: all P·S predictions in one launch).
# The S posteriors of one output share (X, y): `model_posteriors_batched` builds them with ONE `boss_gp_fit_batch` call per output —
# one upload of (X, y), one batched factorisation, S resident handles.
function model_posteriors_batched(m::HipGaussianProcess, ps::AbstractVector{<:HipGPParams}, data::BOSS.ExperimentData)
    X = Matrix{Float64}(data.X); P = size(data.Y, 1); S = length(ps); k = m.gp.kernel
    slices = [Vector{HipPosteriorSlice}(undef, P) for _ in ps]
    for i in 1:P
        y = Vector{Float64}(data.Y[i, :]); mu = BOSS.mean_getindex(m.gp.mean, i); mv = mean_vals(mu, X)
        λ = Matrix{Float64}(reduce(hcat, (p.λ[:, i] for p in ps)))
        hs = fill(C_NULL, S); lp = Vector{Float64}(undef, S); st = zeros(Cint, S)
        check(ccall((:boss_gp_fit_batch, lib), Cint,
            (Cint, Cint, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Ptr{UInt8}, Cint,
             Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Ptr{Cvoid}}, Ptr{Cdouble}, Ptr{Cint}),
            m.device, kernel_id(base_kernel(k)), size(X, 1), size(X, 2), X, y, mv, 0, discrete_flags(k), S,
            λ, Float64[p.α[i] for p in ps], Float64[p.σ[i] for p in ps], hs, lp, st))
        owned = Handle.(hs)                                            # every member is freed by its own finalizer; the shared storage goes with the last
        bad = findfirst(!=(0), st)                                     # a sample that is not PD throws, as `cholesky` would inside the broadcast
        isnothing(bad) || (st[bad] == 3 ? throw(PosDefException(bad)) : error("boss_gp_fit_batch: invalid hyper-parameters in sample $bad"))
        for s in 1:S
            slices[s][i] = HipPosteriorSlice(owned[s], mu)
        end
    end
    return [BOSS.DefaultModelPosterior(sl) for sl in slices]           # what model_posterior(problem) returns under BIParams
end
BOSS.model_posterior(m::HipGaussianProcess, ps::AbstractVector{<:HipGPParams}, data::BOSS.ExperimentData) =
    model_posteriors_batched(m, ps, data)
BOSS.model_posterior(m::HipSemiparametric, ps::AbstractVector{<:HipSemiparametricParams}, data::BOSS.ExperimentData) =
    [BOSS.model_posterior(m, p, data) for p in ps]                     # θ differs per sample: a mean function per sample

"Importance-resampled BI on the device: S candidates from the prior scored in ONE batched likelihood call, `samples` of them drawn
with probability ∝ exp(loglike) — a cheap stand-in where Turing is not loaded; a TuringBI chain works unchanged with these models."
Base.@kwdef struct HipImportanceBI <: BOSS.ModelFitter{BOSS.BIParams}
    candidates::Int
    samples::Int
end
function estimate_parameters(f::HipImportanceBI, problem::BOSS.BossProblem, options::BOSS.BossOptions)
    m = problem.model::Union{HipGaussianProcess, HipSemiparametric}
    sampler = BOSS.params_sampler(m, problem.data)
    ps = [sampler() for _ in 1:f.candidates]                           # proposal = the prior ...
    ll = batched_data_loglike(m, problem.data, ps)                     # ... so the importance weight is the DATA likelihood alone
    w = exp.(ll .- maximum(ll)); w ./= sum(w); cw = cumsum(w)          # (weighting with likelihood × prior would target likelihood × prior²)
    idx = [findfirst(>=(rand()), cw) for _ in 1:f.samples]
    return BOSS.BIParams(samples = [ps[something(i, length(ps))] for i in idx])
end

# ---------------------------------------------------------------- ModelFitter (SamplingMAP semantics, batched)
Base.@kwdef struct HipBatchedMAP <: BOSS.ModelFitter{BOSS.MAPParams}
    samples::Int
    devices::Int = 1              # > 1: the samples split over that many GPUs inside the library (boss_multi_loglike_batch)
end
gp_kernel(m::HipGaussianProcess) = m.gp.kernel
gp_kernel(m::HipSemiparametric) = m.sp.nonparametric.kernel
"Prior means of output i at the data for every parameter sample: (pointer argument, stride) of boss_gp_loglike_batch."
batch_means(m::HipGaussianProcess, ps, X, i) = (mean_vals(BOSS.mean_getindex(m.gp.mean, i), X), 0)      # one vector for all sets
function batch_means(m::HipSemiparametric, ps, X, i)                                                   # S×N, row s = f(x_j; θ_s)[i]
    F = Matrix{Float64}(undef, size(X, 2), length(ps))                 # column s contiguous = row s of the S×N row-major array
    for (s, p) in enumerate(ps)
        f = m.sp.parametric(p.θ)
        F[:, s] .= (f(x)[i] for x in eachcol(X))
    end
    return F, size(X, 2)
end
"Data log-likelihood of every parameter set of `ps` (summed over the outputs): one batched device call per output; -Inf where not PD."
function batched_data_loglike(m::Union{HipGaussianProcess, HipSemiparametric}, data::BOSS.ExperimentData, ps::AbstractVector;
                              devices::Int = 1)
    X = Matrix{Float64}(data.X); S = length(ps); ll = zeros(S); k = gp_kernel(m)
    for i in 1:size(data.Y, 1)
        λ = Matrix{Float64}(reduce(hcat, (p.λ[:, i] for p in ps))); lli = zeros(S); st = zeros(Cint, S)
        mv, stride = batch_means(m, ps, X, i)
        α = Float64[p.α[i] for p in ps]; σ = Float64[p.σ[i] for p in ps]; y = Vector{Float64}(data.Y[i, :])
        if devices > 1                                                  # the S sets split over the GPUs of the node (no collective)
            n = Ref{Cint}(); check(ccall((:boss_init, lib), Cint, (Ref{Cint},), n)); @assert devices <= n[]
            check(ccall((:boss_multi_loglike_batch, lib), Cint,
                (Cint, Cint, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Ptr{UInt8}, Cint,
                 Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cint}),
                devices, kernel_id(base_kernel(k)), size(X, 1), size(X, 2), X, y, mv, stride, discrete_flags(k), S, λ, α, σ, lli, st))
        else
            check(ccall((:boss_gp_loglike_batch, lib), Cint,
                (Cint, Cint, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Ptr{UInt8}, Cint,
                 Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cint}),
                m.device, kernel_id(base_kernel(k)), size(X, 1), size(X, 2), X, y, mv, stride, discrete_flags(k), S, λ, α, σ, lli, st))
        end
        ll .+= lli
    end
    return ll
end
function estimate_parameters(f::HipBatchedMAP, problem::BOSS.BossProblem, options::BOSS.BossOptions; return_all=false)
    m = problem.model::Union{HipGaussianProcess, HipSemiparametric}; data = problem.data
    sampler = BOSS.params_sampler(m, data); prior = params_loglike(m)
    ps = [sampler() for _ in 1:f.samples]
    ll = batched_data_loglike(m, data, ps; devices = f.devices) .+ prior.(ps)     # model_loglike = data + prior (src/surrogate_model.jl)
    return_all && return BOSS.MAPParams.(ps, ll)
    b = argmax(ll); return BOSS.MAPParams(ps[b], ll[b])
end

# ---------------------------------------------------------------- AcquisitionMaximizer (SamplingAM / GridAM semantics, batched)
Base.@kwdef struct HipBatchAM <: BOSS.AcquisitionMaximizer
    x_prior = nothing
    samples::Int = 0
    points::Union{Nothing, Matrix{Float64}} = nothing   # a FIXED candidate set (GridAM's grid, grid.jl:52-65) instead of `samples` draws
    max_attempts::Int = 200
    devices::Int = 1              # > 1: sharded over that many GPUs inside the library (RCCL), along `shard`
    shard::Symbol = :candidates   # :candidates (BASELINE config 3) | :outputs (config 4) | :samples (config 5)
    fused::Bool = true            # one output, one parameter sample: posterior update + acquisition as ONE call (boss_gp_update_acq)
end
"The candidate columns of one acquisition call (sampling.jl:43-46: draws from `x_prior` inside the domain; or the fixed grid)."
function candidates(am::HipBatchAM, problem::BOSS.BossProblem)
    isnothing(am.points) || return am.points
    xs = BOSS._reduce_samples([BOSS._rand_in_domain(am.x_prior, problem.domain; am.max_attempts) for _ in 1:am.samples])
    size(xs, 2) == 0 && @error "HipBatchAM: No samples were successfully drawn!\nCheck the `x_prior` and the `Domain`."
    return Matrix{Float64}(xs)
end
"Prior means at the candidates in the layout boss_acq_ei wants: index p + P*(j + M*s) = a P×M×S array; C_NULL for zero means."
function prior_means(posts::AbstractVector, xs::AbstractMatrix{Float64})
    P = length(posts[1].slices); M = size(xs, 2)
    any(has_mean, posts[1].slices) || return C_NULL
    ms = zeros(P, M, length(posts))
    for (s, post) in enumerate(posts), (p, sl) in enumerate(post.slices)
        isnothing(sl.mean) || (ms[p, :, s] .= mean_vals(sl.mean, xs))      # gaussian_process.jl:101-103: m_p(x_j)
    end
    return ms
end
"What every acquisition entry point takes besides handles and candidates: (coefs, y_max, has_best, best, mask)."
function ei_arguments(problem::BOSS.BossProblem, xs::AbstractMatrix{Float64})
    ei = problem.acquisition::BOSS.ExpectedImprovement{<:BOSS.LinFitness}
    b = BOSS.best_so_far(problem, ei.fitness)
    mask = UInt8[BOSS.in_bounds(x, problem.domain.bounds) && BOSS.in_cons(x, problem.domain.cons) for x in eachcol(xs)]
    ymax = Float64[c for c in problem.y_max]                               # BOSS.Infinity converts to Inf (src/utils/inf.jl)
    return Float64.(ei.fitness.coefs), ymax, isnothing(b) ? Cint(0) : Cint(1), Float64(something(b, 0.0)), ei.cons_safe ? mask : C_NULL
end
"One output, one parameter sample of the plain GP: `model_posterior(problem)` is ONE factorisation, and with candidates that do not
depend on it (sampling.jl:43-57, grid.jl:52-65) the acquisition can ride along: boss_gp_update_acq (bo.jl:30-48 in one device call)."
fusable(am::HipBatchAM, problem::BOSS.BossProblem) = am.fused && am.devices == 1 && problem.model isa HipGaussianProcess &&
    BOSS.y_dim(problem) == 1 && BOSS.get_params(problem) isa HipGPParams
function update_and_acquire(am::HipBatchAM, problem::BOSS.BossProblem, xs::Matrix{Float64}, return_all::Bool)
    m = problem.model::HipGaussianProcess; p = BOSS.get_params(problem)::HipGPParams; data = problem.data
    X = Matrix{Float64}(data.X); y = Vector{Float64}(data.Y[1, :]); M = size(xs, 2); k = m.gp.kernel
    mu = BOSS.mean_getindex(m.gp.mean, 1)
    coefs, ymax, hb, b, mask = ei_arguments(problem, xs)
    h = Ref{Ptr{Cvoid}}(); cand = Ref{Ptr{Cvoid}}()
    check(ccall((:boss_gp_create, lib), Cint, (Cint, Cint, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{UInt8}, Ref{Ptr{Cvoid}}),
          m.device, kernel_id(base_kernel(k)), size(X, 1), size(X, 2), X, y, discrete_flags(k), h))
    gp = Handle(h[])
    check(ccall((:boss_cand_create, lib), Cint, (Cint, Cint, Cint, Ptr{Cdouble}, Ref{Ptr{Cvoid}}), m.device, size(xs, 1), M, xs, cand))
    acq = return_all ? Vector{Float64}(undef, M) : C_NULL
    lp = Ref{Cdouble}(); am_idx = Ref{Clong}(); mx = Ref{Cdouble}(); fused = Ref{Cint}()
    rc = GC.@preserve gp ccall((:boss_gp_update_acq, lib), Cint,
        (Ptr{Cvoid}, Ptr{Cdouble}, Cdouble, Cdouble, Ptr{Cdouble}, Ptr{Cvoid}, Ptr{Cdouble}, Cdouble, Cdouble, Cint, Cdouble, Ptr{UInt8},
         Ref{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Clong}, Ref{Cdouble}, Ref{Cint}),
        gp.h, Vector{Float64}(p.λ[:, 1]), p.α[1], p.σ[1], mean_vals(mu, X), cand[], mean_vals(mu, xs), coefs[1], ymax[1], hb, b, mask,
        lp, C_NULL, C_NULL, acq, am_idx, mx, fused)
    ccall((:boss_cand_free, lib), Cvoid, (Ptr{Cvoid},), cand[])
    check(rc)
    return_all && return xs, acq
    return xs[:, am_idx[] + 1], mx[]
end
function maximize_acquisition(am::HipBatchAM, problem::BOSS.BossProblem, options::BOSS.BossOptions;
                              posts = nothing, return_all::Bool = false)
    xs = candidates(am, problem)
    isnothing(posts) && fusable(am, problem) && return update_and_acquire(am, problem, xs, return_all)
    isnothing(posts) && (posts = BOSS.model_posterior(problem))
    posts isa AbstractVector || (posts = [posts])                          # BI: a vector of posteriors (src/posterior.jl:15-19)
    P = BOSS.y_dim(problem); S = length(posts); M = size(xs, 2)
    hs = Ptr{Cvoid}[posts[s].slices[p].h.h for p in 1:P, s in 1:S]         # P×S, column-major = gps[p + P*s]
    coefs, ymax, hb, b, mask = ei_arguments(problem, xs)
    ms = prior_means(posts, xs)
    acq = return_all ? Vector{Float64}(undef, M) : C_NULL
    am_idx = Ref{Clong}(); mx = Ref{Cdouble}()
    GC.@preserve posts begin
        if am.devices > 1
            rc = multi_acq_ei(am, problem, posts, xs, ms, coefs, ymax, hb, b, mask, acq, am_idx, mx)
        else
            cand = Ref{Ptr{Cvoid}}()
            check(ccall((:boss_cand_create, lib), Cint, (Cint, Cint, Cint, Ptr{Cdouble}, Ref{Ptr{Cvoid}}),
                  problem.model.device, size(xs, 1), M, xs, cand))
            rc = ccall((:boss_acq_ei, lib), Cint,
                (Cint, Cint, Ptr{Ptr{Cvoid}}, Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cdouble, Ptr{UInt8},
                 Ptr{Cdouble}, Ref{Clong}, Ref{Cdouble}),
                P, S, hs, cand[], ms, coefs, ymax, hb, b, mask, acq, am_idx, mx)
            ccall((:boss_cand_free, lib), Cvoid, (Ptr{Cvoid},), cand[])
        end
    end
    check(rc)
    return_all && return xs, acq
    return xs[:, am_idx[] + 1], mx[]
end

# ---- several GPUs from this one process (SURVEY §8e): the three sharding axes of the acquisition
"G replicas of output slice i under hyper-parameters p: created on devices 0..G-1, factorised CONCURRENTLY (boss_multi_gp_update)."
function slice_replicas(G::Int, m::HipGaussianProcess, p::HipGPParams, data::BOSS.ExperimentData, i::Int)
    X = Matrix{Float64}(data.X); y = Vector{Float64}(data.Y[i, :]); k = m.gp.kernel
    mu = BOSS.mean_getindex(m.gp.mean, i)
    hs = map(0:G-1) do g
        h = Ref{Ptr{Cvoid}}()
        check(ccall((:boss_gp_create, lib), Cint, (Cint, Cint, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{UInt8}, Ref{Ptr{Cvoid}}),
              g, kernel_id(base_kernel(k)), size(X, 1), size(X, 2), X, y, discrete_flags(k), h))
        Handle(h[])
    end
    lp = Ref{Cdouble}()
    GC.@preserve hs check(ccall((:boss_multi_gp_update, lib), Cint,
        (Cint, Ptr{Ptr{Cvoid}}, Ptr{Cdouble}, Cdouble, Cdouble, Ptr{Cdouble}, Ref{Cdouble}),
        G, Ptr{Cvoid}[h.h for h in hs], Vector{Float64}(p.λ[:, i]), p.α[i], p.σ[i], mean_vals(mu, X), lp))
    return [HipPosteriorSlice(h, mu) for h in hs]
end
function multi_acq_ei(am::HipBatchAM, problem, posts, xs, ms, coefs, ymax, hb, b, mask, acq, am_idx, mx)
    G = am.devices; nd, _ = init_devices(); @assert G <= nd
    P = BOSS.y_dim(problem); S = length(posts); M = size(xs, 2)
    m = problem.model::HipGaussianProcess
    params = BOSS.get_params(problem.params); params isa AbstractVector || (params = [params])
    if am.shard == :candidates
        # replicas of every posterior on all G devices (concurrent factorisations), the candidate columns split M/G, ONE 16-byte-per-rank
        # all-gather for the arg-max; the shards stay resident for the call (boss_multi_cand_*: reuse them across calls on a fixed grid)
        reps = [slice_replicas(G, m, params[s], problem.data, p) for p in 1:P, s in 1:S]         # [p, s][g]
        hs = Ptr{Cvoid}[reps[p, s][g].h.h for p in 1:P, s in 1:S, g in 1:G]                      # gps[p + P*(s + S*g)]
        mc = Ref{Ptr{Cvoid}}()
        check(ccall((:boss_multi_cand_create, lib), Cint, (Cint, Cint, Cint, Ptr{Cdouble}, Ref{Ptr{Cvoid}}), G, size(xs, 1), M, xs, mc))
        rc = GC.@preserve reps ccall((:boss_multi_acq_ei_cand, lib), Cint,
            (Cint, Cint, Cint, Ptr{Ptr{Cvoid}}, Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cdouble, Ptr{UInt8},
             Ptr{Cdouble}, Ref{Clong}, Ref{Cdouble}),
            G, P, S, hs, mc[], ms, coefs, ymax, hb, b, mask, acq, am_idx, mx)
        ccall((:boss_multi_cand_free, lib), Cvoid, (Ptr{Cvoid},), mc[])
        return rc
    end
    # :outputs — slice p lives on device (p-1) % G;  :samples — all outputs of sample s live on device (s-1) % G.  Each posterior is
    # fitted where it lives; (μ, σ²) rows resp. partial acquisition sums travel in ONE all-reduce (host_multi.inc)
    own(p, s) = am.shard == :outputs ? (p - 1) % G : (s - 1) % G
    sl = [model_posterior_slice(HipGaussianProcess(m.gp, Cint(own(p, s))), params[s], problem.data, p) for p in 1:P, s in 1:S]
    hs = Ptr{Cvoid}[sl[p, s].h.h for p in 1:P, s in 1:S]
    if am.shard == :outputs
        return GC.@preserve sl ccall((:boss_multi_acq_ei_outputs, lib), Cint,
            (Cint, Cint, Ptr{Ptr{Cvoid}}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cdouble, Ptr{UInt8},
             Ptr{Cdouble}, Ref{Clong}, Ref{Cdouble}),
            P, S, hs, M, xs, ms, coefs, ymax, hb, b, mask, acq, am_idx, mx)
    end
    return GC.@preserve sl ccall((:boss_multi_acq_ei_samples, lib), Cint,
        (Cint, Cint, Ptr{Ptr{Cvoid}}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cdouble, Ptr{UInt8},
         Ptr{Cdouble}, Ref{Clong}, Ref{Cdouble}),
        P, S, hs, M, xs, ms, coefs, ymax, hb, b, mask, acq, am_idx, mx)
end
# ---------------------------------------------------------------- SequentialBatchAM on resident posteriors
# batch.jl:26-38 rebuilds the posterior (an O(N^3) Cholesky) for every speculative point; here the
# handles stay resident and each speculative observation is a block Cholesky append (O(N^2)).
function append!(post::HipPosteriorSlice, x::AbstractVector{<:Real}, y::Real)
    lp = Ref{Cdouble}(); X = reshape(Vector{Float64}(x), :, 1)
    check(ccall((:boss_gp_append, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Cdouble}),
          post.h.h, 1, X, Float64[y], mean_vals(post.mean, X), lp))    # mean_new is required when the posterior has a prior mean
    return lp[]
end
"Observations currently in the handle (`size(data.X, 2)` after augment_dataset!, also after an append that failed part-way)."
n_obs(post::HipPosteriorSlice) = (n = Ref{Cint}(); check(ccall((:boss_gp_n, lib), Cint, (Ptr{Cvoid}, Ref{Cint}), post.h.h, n)); Int(n[]))
"Room for `extra` later appends without re-allocation; leaves the handle unfitted (follow with an update under the same parameters)."
reserve!(post::HipPosteriorSlice, extra::Int) =
    check(ccall((:boss_gp_reserve, lib), Cint, (Ptr{Cvoid}, Cint), post.h.h, n_obs(post) + extra))
"New observations of the same inputs (a re-evaluated objective): y only, the points stay resident; follow with an update."
set_y!(post::HipPosteriorSlice, y::AbstractVector{<:Real}) =
    check(ccall((:boss_gp_set_y, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), post.h.h, Vector{Float64}(y)))
Base.@kwdef struct HipSequentialBatchAM <: BOSS.AcquisitionMaximizer
    am::HipBatchAM
    batch_size::Int
end
function maximize_acquisition(sb::HipSequentialBatchAM, problem::BOSS.BossProblem, options::BOSS.BossOptions)
    problem_ = deepcopy(problem); post = BOSS.model_posterior(problem_)
    post isa AbstractVector && return sequential_untracked(sb, problem_, post, options)       # BI: appends to every sample's posterior
    isnothing(sb.am.points) || sb.am.devices > 1 || return sequential_tracked(sb, problem_, post)
    return sequential_untracked(sb, problem_, [post], options)
end
function sequential_untracked(sb, problem_, posts, options)
    X = reduce(hcat, map(1:sb.batch_size) do _
        x, _ = maximize_acquisition(sb.am, problem_, options; posts)          # HipBatchAM with given handles
        y = sum(BOSS.mean(post, x) for post in posts) ./ length(posts)      # mean(post, x); BI: average_mean (src/posterior.jl:177-179)
        BOSS.augment_dataset!(problem_, x, y)
        foreach(post -> foreach(i -> append!(post.slices[i], x, y[i]), eachindex(y)), posts)
        x
    end)
    return X, nothing
end
"Fixed candidate set (`points`): the candidates' V = C.U' \\ K* slabs stay resident (boss_track_*) and every speculative observation
extends them by ONE row — O(N·M) per selection instead of the O(N²M) re-solve (batch.jl:32-38)."
function sequential_tracked(sb, problem_, post)
    xs = sb.am.points; P = BOSS.y_dim(problem_); M = size(xs, 2)
    cand = Ref{Ptr{Cvoid}}()
    check(ccall((:boss_cand_create, lib), Cint, (Cint, Cint, Cint, Ptr{Cdouble}, Ref{Ptr{Cvoid}}), problem_.model.device, size(xs, 1), M, xs, cand))
    tracks = map(post.slices) do sl
        t = Ref{Ptr{Cvoid}}()
        check(ccall((:boss_track_create, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cdouble}, Ref{Ptr{Cvoid}}), sl.h.h, cand[], mean_vals(sl.mean, xs), t))
        t[]
    end
    X = try
        reduce(hcat, map(1:sb.batch_size) do _
            coefs, ymax, hb, b, mask = ei_arguments(problem_, xs)           # best_so_far moves with the speculative data
            am_idx = Ref{Clong}(); mx = Ref{Cdouble}()
            check(ccall((:boss_acq_ei_tracks, lib), Cint,
                (Cint, Cint, Ptr{Ptr{Cvoid}}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cdouble, Ptr{UInt8}, Ptr{Cdouble}, Ref{Clong}, Ref{Cdouble}),
                P, 1, tracks, coefs, ymax, hb, b, mask, C_NULL, am_idx, mx))
            x = xs[:, am_idx[] + 1]
            y = map(tracks) do t                                           # ŷ = mean(post, x): read off the tracked state
                μ = Ref{Cdouble}(); σ2 = Ref{Cdouble}()
                check(ccall((:boss_track_moments, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Ref{Cdouble}, Ref{Cdouble}), t, am_idx[], 1, μ, σ2))
                μ[]
            end
            BOSS.augment_dataset!(problem_, x, y)
            foreach(i -> append!(post.slices[i], x, y[i]), eachindex(y))
            foreach(t -> check(ccall((:boss_track_sync, lib), Cint, (Ptr{Cvoid},), t)), tracks)   # one new row of V per track
            x
        end)
    finally
        foreach(t -> ccall((:boss_track_free, lib), Cvoid, (Ptr{Cvoid},), t), tracks)
        ccall((:boss_cand_free, lib), Cvoid, (Ptr{Cvoid},), cand[])
    end
    return X, nothing
end
# ---------------------------------------------------------------- analytic gradients (instead of ForwardDiff duals)
# value and gradient of the acquisition at the columns of X (d×M), one device call for all columns:
function acq_value_and_grad(problem::BOSS.BossProblem, post, X::AbstractMatrix{<:Real})
    ei = problem.acquisition::BOSS.ExpectedImprovement{<:BOSS.LinFitness}
    P = BOSS.y_dim(problem); Xs = Matrix{Float64}(X); d, M = size(Xs)
    hs = Ptr{Cvoid}[post.slices[p].h.h for p in 1:P]
    b = BOSS.best_so_far(problem, ei.fitness)
    mask = UInt8[BOSS.in_bounds(x, problem.domain.bounds) && BOSS.in_cons(x, problem.domain.cons) for x in eachcol(Xs)]
    # prior means [p][M] (an M×P matrix) and their gradients [p][d×M] (a d×M×P array); constant means have zero gradient,
    # function means are differentiated with ForwardDiff (a BOSS dependency) — only the MEAN, not the GP, goes through duals
    ms, mg = C_NULL, C_NULL
    if any(has_mean, post.slices)
        ms = zeros(M, P); mg = zeros(d, M, P)
        for (p, sl) in enumerate(post.slices)
            isnothing(sl.mean) && continue
            ms[:, p] .= mean_vals(sl.mean, Xs)
            sl.mean isa Function && (mg[:, :, p] .= reduce(hcat, (BOSS.ForwardDiff.gradient(sl.mean, Vector(x)) for x in eachcol(Xs))))
        end
    end
    acq = Vector{Float64}(undef, M); dacq = Matrix{Float64}(undef, d, M)
    GC.@preserve post check(ccall((:boss_acq_ei_grad, lib), Cint,
        (Cint, Ptr{Ptr{Cvoid}}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cdouble,
         Ptr{UInt8}, Ptr{Cdouble}, Ptr{Cdouble}),
        P, hs, M, Xs, ms, mg, Float64.(ei.fitness.coefs), Float64[c for c in problem.y_max], isnothing(b) ? 0 : 1,
        something(b, 0.0), ei.cons_safe ? mask : C_NULL, acq, dacq))
    return acq, dacq          # feed an Optimization.jl OptimizationFunction(f; grad = ...) per start, or batch the starts
end
# ---------------------------------------------------------------- likelihood gradient for OptimizationMAP-style fitters
# (value, gradient) of the data log-likelihood of output slice i at hyper-parameters p, on resident data:
function loglike_and_grad!(h::Ptr{Cvoid}, λ::Vector{Float64}, α::Float64, σ::Float64, mean_X)
    lp = Ref{Cdouble}(); g = Vector{Float64}(undef, length(λ) + 2)
    check(ccall((:boss_gp_update, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cdouble, Cdouble, Ptr{Cdouble}, Cint, Ref{Cdouble}),
          h, λ, α, σ, mean_X, 0, lp))
    check(ccall((:boss_gp_loglike_grad, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}), h, C_NULL, g))
    return lp[], g            # g = [∂/∂λ_1 … ∂/∂λ_d, ∂/∂α, ∂/∂σ]; add the priors' gradients and chain through the bijector
end
# all starts of a multistart fit per round: values and gradients of S hyper-parameter sets of output slice i in one call
function loglike_and_grad_batch(m::HipGaussianProcess, data::BOSS.ExperimentData, i::Int, λ::Matrix{Float64} #= d×S =#,
                                α::Vector{Float64}, σ::Vector{Float64})
    X = Matrix{Float64}(data.X); d, N = size(X); S = length(α)
    ll = Vector{Float64}(undef, S); g = Matrix{Float64}(undef, d + 2, S); st = Vector{Cint}(undef, S)
    mu = BOSS.mean_getindex(m.gp.mean, i)
    check(ccall((:boss_gp_loglike_grad_batch, lib), Cint,
        (Cint, Cint, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Ptr{UInt8}, Cint, Ptr{Cdouble}, Ptr{Cdouble},
         Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cint}),
        m.device, kernel_id(base_kernel(m.gp.kernel)), d, N, X, Vector{Float64}(data.Y[i, :]), mean_vals(mu, X), 0,
        discrete_flags(m.gp.kernel), S, λ, α, σ, ll, g, st))
    return ll, g, st          # ll[s] = -Inf and g[:, s] = 0 where st[s] != 0 (not PD / invalid parameters)
end
# ---------------------------------------------------------------- OptimizationAM semantics on analytic device gradients
"""
`HipGradientAM` — OptimizationAM semantics (src/acquisition_maximizers/optimization.jl:13-118): multistart LOCAL optimisation of
the acquisition from `multistart` starts inside the domain, the best local optimum wins (`optimize_multistart`,
src/utils/optim_multistart.jl).  The reference differentiates the acquisition with ForwardDiff (:36) one start at a time; here
the gradient is analytic and evaluated on the device for ALL starts in one call per iteration (`boss_acq_ei_grad`): projected
gradient ascent with a per-start step and backtracking.  Discrete dimensions are rounded (gradient 0); `cons` is honoured through
make_safe (acq = 0 outside) and a final in-domain filter.  Works on `HipGaussianProcess` / `HipSemiparametric` posteriors and on those of
`HipGradientGaussianProcess` (`boss_acq_ei_grad` differentiates the augmented cross-covariances of src/models/gradient_gp.jl:221-243 as well).
"""
Base.@kwdef struct HipGradientAM <: BOSS.AcquisitionMaximizer
    multistart::Union{Int, Matrix{Float64}} = 200        # number of starts, or the starts as columns (set_starts, optimization.jl:43-52)
    iters::Int = 30
end
BOSS.set_starts(am::HipGradientAM, starts::AbstractMatrix{<:Real}) = HipGradientAM(Matrix{Float64}(starts), am.iters)
function maximize_acquisition(am::HipGradientAM, problem::BOSS.BossProblem, options::BOSS.BossOptions;
                              posts = BOSS.model_posterior(problem))
    dom = problem.domain; lb, ub = Float64.(dom.bounds[1]), Float64.(dom.bounds[2])
    X = Matrix{Float64}(BOSS.get_starts(am.multistart, dom))              # LHC starts (optimization.jl:78-88) or the given ones
    posts isa AbstractVector || (posts = [posts])
    value_and_grad(Z) = begin                                              # BI: the sample mean of acquisition and gradient (:87-90)
        vg = [acq_value_and_grad(problem, post, Z) for post in posts]
        sum(first.(vg)) ./ length(posts), sum(last.(vg)) ./ length(posts)
    end
    span = map((l, u) -> isfinite(u - l) ? u - l : 1.0, lb, ub); cont = .!dom.discrete
    f, g = value_and_grad(X); step = fill(0.05, size(X, 2))                # step relative to the box, per start
    for _ in 1:am.iters
        gn = g .* span; gn[.!cont, :] .= 0.0
        nrm = max.(sqrt.(vec(sum(abs2, gn; dims = 1))), 1e-300)
        Xn = clamp.(X .+ (gn ./ nrm') .* span .* step', lb, ub)
        fn, gnw = value_and_grad(Xn)
        better = fn .> f
        X[:, better] .= Xn[:, better]; f[better] .= fn[better]; g[:, better] .= gnw[:, better]
        step .= ifelse.(better, min.(step .* 1.5, 0.5), step .* 0.4)
        all(<(1e-7), step) && break
    end
    X .= BOSS.cond_func(round).(X, dom.discrete)                           # assure discrete dims (optimization.jl:115)
    ok = [BOSS.in_domain(x, dom) for x in eachcol(X)]
    f, _ = value_and_grad(X); f[.!ok] .= -Inf
    j = argmax(f); return X[:, j], f[j]
end

# ---------------------------------------------------------------- OptimizationMAP semantics on analytic device gradients
"""
`HipGradientMAP` — OptimizationMAP semantics (src/model_fitters/optimization.jl:13-164): multistart local maximisation of the
log-posterior `loglike(data | θ) + logprior(θ)` over the GP hyper-parameters, the best local optimum wins.  The reference hands
the objective to an Optimization.jl algorithm with automatic differentiation (:146-164); here the starts advance in lockstep and
every round is ONE device call per output — `boss_gp_loglike_grad_batch`: values and analytic gradients of all trial points —
and the ascent runs in log-parameter space (positive parameters; the reference's bijector maps them the same way).  Parameters
with a Dirac prior stay fixed (src/models/utils/dirac.jl:36-77); the priors' gradient comes from ForwardDiff (a few scalars).
"""
Base.@kwdef struct HipGradientMAP <: BOSS.ModelFitter{BOSS.MAPParams}
    multistart::Union{Int, Vector{<:HipGPParams}} = 8
    iters::Int = 40
    step0::Float64 = 0.3
end
BOSS.set_starts(f::HipGradientMAP, starts::AbstractVector{<:HipGPParams}) = HipGradientMAP(collect(starts), f.iters, f.step0)
flat(p::HipGPParams) = vcat(vec(p.λ), p.α, p.σ)
unflat(p::HipGPParams, v::AbstractVector) = (n = length(p.λ); P = length(p.α);
    HipGPParams(reshape(v[1:n], size(p.λ)), v[n+1:n+P], v[n+P+1:n+2P]))
"log-posterior and its gradient w.r.t. flat(p) = [vec(λ); α; σ] for every parameter set: one batched device call per output."
function objective_batch(m::HipGaussianProcess, data::BOSS.ExperimentData, ps::AbstractVector{<:HipGPParams})
    prior = params_loglike(m); S = length(ps); d, P = size(ps[1].λ)
    f = Float64[prior(p) for p in ps]
    G = [BOSS.ForwardDiff.gradient(v -> prior(unflat(p, v)), flat(p)) for p in ps]      # the priors: host scalars
    for i in 1:P
        λ = Matrix{Float64}(reduce(hcat, (p.λ[:, i] for p in ps)))
        ll, g, st = loglike_and_grad_batch(m, data, i, λ, Float64[p.α[i] for p in ps], Float64[p.σ[i] for p in ps])
        for s in 1:S
            f[s] += st[s] == 0 ? ll[s] : -Inf
            G[s][(i-1)*d+1:i*d] .+= g[1:d, s]; G[s][d*P+i] += g[d+1, s]; G[s][d*P+P+i] += g[d+2, s]
        end
    end
    return f, G
end
function estimate_parameters(fit::HipGradientMAP, problem::BOSS.BossProblem, options::BOSS.BossOptions; return_all::Bool = false)
    m = problem.model::HipGaussianProcess; data = problem.data
    sampler = BOSS.params_sampler(m, data)
    ps = fit.multistart isa Int ? [sampler() for _ in 1:fit.multistart] : copy(fit.multistart)
    free = .!first(BOSS.create_dirac_mask(BOSS.param_priors(m)))            # over flat(p): Dirac-prior parameters do not move
    f, G = objective_batch(m, data, ps)
    step = fill(fit.step0, length(ps)); its = zeros(Int, length(ps)); alive = isfinite.(f) .& (fit.iters > 0)
    while any(alive)
        idx = Int[]; trial = eltype(ps)[]
        for k in findall(alive)
            v = flat(ps[k]); dir = ifelse.(free, v .* G[k], 0.0)            # ∂f/∂log θ = θ ∂f/∂θ
            nrm = sqrt(sum(abs2, dir))
            nrm < 1e-10 && (alive[k] = false; continue)
            push!(idx, k); push!(trial, unflat(ps[k], v .* exp.(step[k] .* dir ./ nrm)))
        end
        isempty(idx) && break
        fq, Gq = objective_batch(m, data, trial)                            # ALL trial points of the round in one call per output
        for (j, k) in enumerate(idx)
            if fq[j] > f[k]
                ps[k], f[k], G[k] = trial[j], fq[j], Gq[j]
                step[k] = min(step[k] * 1.6, 2.0); its[k] += 1
                its[k] >= fit.iters && (alive[k] = false)
            else
                step[k] *= 0.4
                step[k] <= 1e-6 && (alive[k] = false)
            end
        end
    end
    return_all && return BOSS.MAPParams.(ps, f)
    b = argmax(f); return BOSS.MAPParams(ps[b], f[b])
end

# ---------------------------------------------------------------- GradientGaussianProcess (values + gradients)
# BOSS.GradientGaussianProcess with the n(1+d) augmented system on the device; data::BOSS.GradientData.
struct HipGradientGaussianProcess{G<:BOSS.GradientGaussianProcess} <: BOSS.SurrogateModel
    gp::G
    device::Cint
end
for f in (:params_loglike, :_params_sampler, :vectorizer, :bijector, :param_lengths)
    @eval BOSS.$f(m::HipGradientGaussianProcess, args...) = BOSS.$f(m.gp, args...)
end
sliceable(::HipGradientGaussianProcess) = true
slice(m::HipGradientGaussianProcess, i::Int) = HipGradientGaussianProcess(slice(m.gp, i), m.device)
function ggp_create(m::HipGradientGaussianProcess, data::BOSS.GradientData, i::Int)
    X = Matrix{Float64}(data.X); h = Ref{Ptr{Cvoid}}()
    dY = ndims(data.dY) == 3 ? Matrix{Float64}(data.dY[i, :, :]) : Matrix{Float64}(data.dY)     # x_dim × n
    check(ccall((:boss_ggp_create, lib), Cint, (Cint, Cint, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Ptr{Cvoid}}),
          m.device, kernel_id(m.gp.kernel), size(X, 1), size(X, 2), X, Vector{Float64}(data.Y[i, :]), dY, h))
    return h[]
end
ggp_update(h, p::BOSS.GradientGaussianProcessParams, i::Int) = (lp = Ref{Cdouble}();
    check(ccall((:boss_ggp_update, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cdouble, Cdouble, Cdouble, Cint, Ref{Cdouble}),
          h, Vector{Float64}(p.λ[:, i]), p.α[i], p.σ[i], p.σ_∂[i], 0, lp)); lp[])
function model_posterior_slice(m::HipGradientGaussianProcess, p::BOSS.GradientGaussianProcessParams, data::BOSS.GradientData, i::Int)
    h = ggp_create(m, data, i); ggp_update(h, p, i)
    return HipPosteriorSlice(Handle(h), nothing)  # mean / var / mean_and_var above apply (gradient_gp.jl:334-361)
end
function data_loglike(m::HipGradientGaussianProcess, data::BOSS.GradientData)
    h = Handle(ggp_create(m, data, 1))            # per-output likelihood of the sliced model (gradient_gp.jl:367-397)
    return p -> try ggp_update(h.h, p, 1) catch e; e isa PosDefException ? -Inf : rethrow() end
end
"""
data_loglike with its gradient, for gradient-based fitters (what ForwardDiff yields through gradient_gp.jl:367-397 inside
OptimizationMAP, src/model_fitters/optimization.jl:146-164): p -> (ℓ, (∂ℓ/∂λ (x_dim), ∂ℓ/∂α, ∂ℓ/∂σ, ∂ℓ/∂σ_∂)) of the sliced model.
"""
function data_loglike_grad(m::HipGradientGaussianProcess, data::BOSS.GradientData)
    h = Handle(ggp_create(m, data, 1)); d = size(data.X, 1)
    return function (p)
        ggp_update(h.h, p, 1)
        lp = Ref{Cdouble}(); g = Vector{Float64}(undef, d + 3)
        GC.@preserve h check(ccall((:boss_ggp_loglike_grad, lib), Cint, (Ptr{Cvoid}, Ref{Cdouble}, Ptr{Cdouble}), h.h, lp, g))
        return lp[], (g[1:d], g[d + 1], g[d + 2], g[d + 3])
    end
end
"augment_dataset! (src/types/problem.jl:191-198) for a fitted gradient-observation slice: new points with values and gradients, same hyper-parameters."
function augment!(post::HipPosteriorSlice, X_new::AbstractMatrix{<:Real}, y_new::AbstractVector{<:Real}, dY_new::AbstractMatrix{<:Real})
    lp = Ref{Cdouble}()
    GC.@preserve post check(ccall((:boss_ggp_append, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Cdouble}),
        post.h.h, size(X_new, 2), Matrix{Float64}(X_new), Vector{Float64}(y_new), Matrix{Float64}(dY_new), lp))
    return lp[]
end
# ---------------------------------------------------------------- NonstationaryGP (Gibbs kernel)
# The latent models stay BOSS's own (ParametrizedGP posteriors or constants); only their values cross the ABI.
struct HipNonstationaryPosterior <: BOSS.ModelPosteriorSlice{BOSS.NonstationaryGP}
    post::HipPosteriorSlice; f_λ; f_α; discrete; f_σ
end
rounded(X, ::Nothing) = X
rounded(X, disc) = (Xr = copy(X); Xr[disc, :] .= round.(Xr[disc, :]); Xr)
function hip_posterior_slice(model::BOSS.NonstationaryGP, params::BOSS.NonstationaryGPParams, data::BOSS.ExperimentData,
                             i::Int; device = 0)
    f_λ = BOSS._param_posterior_slice(model.lengthscale_model, params.λ, data, i)
    f_α = BOSS._param_posterior_slice(model.amplitude_model, params.α, data, i)
    f_σ = BOSS._param_posterior_slice(model.noise_std_model, params.σ, data, i)
    X = Matrix{Float64}(data.X); Xr = rounded(X, model.discrete); mu = BOSS.mean_getindex(model.mean, i)
    h = Ref{Ptr{Cvoid}}(); lp = Ref{Cdouble}()
    check(ccall((:boss_ngp_create, lib), Cint, (Cint, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{UInt8}, Ref{Ptr{Cvoid}}),
          device, size(X, 1), size(X, 2), X, Vector{Float64}(data.Y[i, :]), isnothing(model.discrete) ? C_NULL : UInt8.(model.discrete), h))
    check(ccall((:boss_ngp_update, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Ref{Cdouble}),
          h[], reduce(hcat, f_λ.(eachcol(Xr))), Float64.(f_α.(eachcol(Xr))), Float64.(f_σ.(eachcol(X))), mean_vals(mu, X), 0, lp))
    return HipNonstationaryPosterior(HipPosteriorSlice(Handle(h[]), mu), f_λ, f_α, model.discrete, f_σ), lp[]   # lp = data_loglike_slice
end
"""
Log-likelihood of a fitted nonstationary slice with its partial derivatives w.r.t. the latent models' values at the training points:
(ℓ, ∂ℓ/∂λ(x_j) (x_dim × N), ∂ℓ/∂α(x_j), ∂ℓ/∂σ(x_j), ∂ℓ/∂m(x_j)) — the cotangents a reverse rule for `data_loglike_slice`
(nonstationary_gp.jl:237-245) hands to the latent `ParametrizedGP` posteriors; with ForwardDiff duals in the latent parameters the
directional derivative is their dot product with the duals' partials.
"""
function loglike_grad_values(p::HipNonstationaryPosterior, N::Int, d::Int)
    lp = Ref{Cdouble}(); dλ = Matrix{Float64}(undef, d, N); dα = Vector{Float64}(undef, N); dσ = similar(dα); dm = similar(dα)
    GC.@preserve p check(ccall((:boss_ngp_loglike_grad, lib), Cint, (Ptr{Cvoid}, Ref{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}),
        p.post.h.h, lp, dλ, dα, dσ, dm))
    return lp[], dλ, dα, dσ, dm
end
"augment_dataset! (src/types/problem.jl:191-198) for a fitted nonstationary slice: the latent models are evaluated at the new points only."
function augment!(p::HipNonstationaryPosterior, X_new::AbstractMatrix{<:Real}, y_new::AbstractVector{<:Real})
    Xn = Matrix{Float64}(X_new); Xr = rounded(Xn, p.discrete); lp = Ref{Cdouble}()
    GC.@preserve p check(ccall((:boss_ngp_append, lib), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Cdouble}),
        p.post.h.h, size(Xn, 2), Xn, Vector{Float64}(y_new), reduce(hcat, p.f_λ.(eachcol(Xr))), Float64.(p.f_α.(eachcol(Xr))),
        Float64.(p.f_σ.(eachcol(Xn))), mean_vals(p.post.mean, Xn), lp))
    return lp[]
end
function mean_and_var(p::HipNonstationaryPosterior, X::AbstractMatrix{<:Real})
    Xs = Matrix{Float64}(X); Xr = rounded(Xs, p.discrete); M = size(Xs, 2)
    μ = Vector{Float64}(undef, M); σ2 = similar(μ); bad = Ref{Clong}(-1)
    check(ccall((:boss_ngp_predict, lib), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Clong}),
        p.post.h.h, M, Xs, reduce(hcat, p.f_λ.(eachcol(Xr))), Float64.(p.f_α.(eachcol(Xr))), mean_vals(p.post.mean, Xs), μ, σ2, bad))
    return μ, σ2
end
# ---------------------------------------------------------------- moments with gradients, moments-only acquisition
"mean_and_var(post, X) with their analytic gradients w.r.t. the columns of X: (μ, σ², ∂μ/∂x (d×M), ∂σ²/∂x (d×M))."
function mean_and_var_grad(post::HipPosteriorSlice, X::AbstractMatrix{<:Real}; mean_grad = C_NULL)
    Xs = Matrix{Float64}(X); d, M = size(Xs)
    μ = Vector{Float64}(undef, M); σ2 = similar(μ); dμ = Matrix{Float64}(undef, d, M); dσ2 = similar(dμ); bad = Ref{Clong}(-1)
    GC.@preserve post check(ccall((:boss_gp_predict_grad, lib), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Clong}),
        post.h.h, M, Xs, mean_vals(post.mean, Xs), mean_grad, μ, σ2, dμ, dσ2, bad))
    return μ, σ2, dμ, dσ2
end
"""
The same for a nonstationary posterior: the candidate enters the Gibbs kernel directly and through λ(x*), α(x*); the Jacobians of the
latent models (BOSS's own ParametrizedGP posteriors or constants, host closures) come from ForwardDiff — what OptimizationAM's AD
(src/acquisition_maximizers/optimization.jl:36) would push through them — and everything over the N observations runs on the device.
"""
function mean_and_var_grad(p::HipNonstationaryPosterior, X::AbstractMatrix{<:Real}; mean_grad = C_NULL)
    Xs = Matrix{Float64}(X); Xr = rounded(Xs, p.discrete); d, M = size(Xs)
    λ = reduce(hcat, p.f_λ.(eachcol(Xr))); α = Float64.(p.f_α.(eachcol(Xr)))
    Dλ = Array{Float64}(undef, d, d, M); Dα = Matrix{Float64}(undef, d, M)
    for (j, x) in enumerate(eachcol(Xr))
        Dλ[:, :, j] .= ForwardDiff.jacobian(p.f_λ, collect(x)); Dα[:, j] .= ForwardDiff.gradient(p.f_α, collect(x))
    end
    isnothing(p.discrete) || (Dλ[:, p.discrete, :] .= 0.0; Dα[p.discrete, :] .= 0.0)      # rounded dimensions: piecewise constant
    μ = Vector{Float64}(undef, M); σ2 = similar(μ); dμ = Matrix{Float64}(undef, d, M); dσ2 = similar(dμ); bad = Ref{Clong}(-1)
    check(ccall((:boss_ngp_predict_grad, lib), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble},
         Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Clong}),
        p.post.h.h, M, Xs, λ, α, Dλ, Dα, mean_vals(p.post.mean, Xs), mean_grad, μ, σ2, dμ, dσ2, bad))
    return μ, σ2, dμ, dσ2
end
"EI × feasibility and its gradient w.r.t. the candidates from per-output moments and gradients (P-vectors of what `mean_and_var_grad` returns)."
function acq_grad_from_moments(problem::BOSS.BossProblem, xs::AbstractMatrix{Float64}, moms::AbstractVector; device = 0)
    d, M = size(xs); P = length(moms); coefs, ymax, hb, b, mask = ei_arguments(problem, xs)
    mu = reduce(hcat, (m[1] for m in moms)); var = reduce(hcat, (max.(m[2], 0.0) for m in moms))      # M×P = [p][M]
    dmu = cat((m[3] for m in moms)...; dims = 3); dvar = cat((m[4] for m in moms)...; dims = 3)        # d×M×P = [p][d×M]
    acq = Vector{Float64}(undef, M); dacq = Matrix{Float64}(undef, d, M)
    check(ccall((:boss_acq_ei_grad_moments, lib), Cint,
        (Cint, Cint, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cdouble,
         Ptr{UInt8}, Ptr{Cdouble}, Ptr{Cdouble}),
        device, P, M, d, mu, var, dmu, dvar, coefs, ymax, hb, b, mask, acq, dacq))
    return acq, dacq
end
"EI × feasibility and arg-max from moments the caller already holds (e.g. of a HipNonstationaryPosterior): mu, var are M×P×S arrays."
function acq_from_moments(problem::BOSS.BossProblem, xs::AbstractMatrix{Float64}, mu::Array{Float64, 3}, var::Array{Float64, 3}; device = 0)
    M, P, S = size(mu); coefs, ymax, hb, b, mask = ei_arguments(problem, xs)
    acq = Vector{Float64}(undef, M); am_idx = Ref{Clong}(); mx = Ref{Cdouble}()
    check(ccall((:boss_acq_ei_moments, lib), Cint,
        (Cint, Cint, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cdouble, Ptr{UInt8}, Ptr{Cdouble}, Ref{Clong}, Ref{Cdouble}),
        device, P, S, M, mu, var, coefs, ymax, hb, b, mask, acq, am_idx, mx))
    return acq, am_idx[] + 1, mx[]
end

# ---------------------------------------------------------------- entry points of include/bosship.h this glue does not call
# (tests/test_abi_and_host.py checks that every exported symbol is either bound above or listed here with its reason)
# not bound: boss_set_stream — runs the library on a caller's HIP stream (torch / AMDGPU.jl interop); BOSS.jl itself owns no stream
# not bound: boss_device_sync — drains that stream; every entry point used above returns synchronised results
# not bound: boss_gp_get_factor — test introspection (L and z of a fitted handle); no BOSS.jl API asks for the factor
# not bound: boss_bench_mfma_f64 — measurement helper of bench.py (fp64 MFMA issue-rate probe)
# not bound: boss_prof_enable — measurement helper of bench.py (per-kernel-class HIP-event timers)
# not bound: boss_prof_reset — measurement helper of bench.py
# not bound: boss_prof_get — measurement helper of bench.py
# not bound: boss_multi_acq_ei — superseded here by boss_multi_cand_create + boss_multi_acq_ei_cand (same exchange, candidate shards uploaded once)
end # module
