module BOSSHip
using BOSS, LinearAlgebra
import BOSS: model_posterior_slice, data_loglike, params_loglike, _params_sampler, vectorizer, bijector,
             sliceable, slice, join_slices, make_discrete, mean, var, mean_and_var,
             estimate_parameters, maximize_acquisition

const lib = joinpath(@__DIR__, "libbosship.so")
check(rc) = rc == 0 ? nothing :
    rc == 3 ? throw(PosDefException(0)) :
    rc == 4 ? throw(DomainError(NaN, unsafe_string(ccall((:boss_last_error, lib), Cstring, ())))) :
              error(unsafe_string(ccall((:boss_last_error, lib), Cstring, ())))
kernel_id(k) = k isa BOSS.Matern32Kernel ? 0 : k isa BOSS.Matern52Kernel ? 1 : 2      # SqExponential / Gaussian

# ---------------------------------------------------------------- SurrogateModel
"GaussianProcess whose posterior lives on an MI355X. Wraps a BOSS.GaussianProcess for priors/vectorizer/bijector."
struct HipGaussianProcess{G<:BOSS.GaussianProcess} <: BOSS.SurrogateModel
    gp::G
    device::Cint
end
const HipGPParams = BOSS.GaussianProcessParams         # same (λ, α, σ) container
for f in (:params_loglike, :_params_sampler, :vectorizer, :bijector)
    @eval $f(m::HipGaussianProcess, args...) = $f(m.gp, args...)
end
sliceable(::HipGaussianProcess) = true
slice(m::HipGaussianProcess, i::Int) = HipGaussianProcess(slice(m.gp, i), m.device)
make_discrete(m::HipGaussianProcess, d::AbstractVector{Bool}) = HipGaussianProcess(make_discrete(m.gp, d), m.device)

mutable struct HipPosteriorSlice <: BOSS.ModelPosteriorSlice{HipGaussianProcess}
    h::Ptr{Cvoid}; mean                                            # prior mean closure (or nothing)
    function HipPosteriorSlice(h, mean); p = new(h, mean)
        finalizer(p -> ccall((:boss_gp_free, lib), Cvoid, (Ptr{Cvoid},), p.h), p); end
end
mean_vals(::Nothing, X) = C_NULL
mean_vals(m::Real, X) = fill(Float64(m), size(X, 2))
mean_vals(m::Function, X) = Float64[m(x) for x in eachcol(X)]
discrete_flags(k) = k isa BOSS.DiscreteKernel && !(k.dims isa Missing) ? UInt8.(k.dims) : C_NULL

function model_posterior_slice(m::HipGaussianProcess, p::HipGPParams, data::BOSS.ExperimentData, i::Int)
    X = Matrix{Float64}(data.X); y = Vector{Float64}(data.Y[i, :])
    mu = BOSS.mean_getindex(m.gp.mean, i)
    h = Ref{Ptr{Cvoid}}(); lp = Ref{Cdouble}()
    check(ccall((:boss_gp_fit, lib), Cint,
        (Cint, Cint, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Cdouble,
         Ptr{UInt8}, Ref{Ptr{Cvoid}}, Ref{Cdouble}),
        m.device, kernel_id(m.gp.kernel), size(X, 1), size(X, 2), X, y, mean_vals(mu, X),
        Vector{Float64}(p.λ[:, i]), p.α[i], p.σ[i], discrete_flags(m.gp.kernel), h, lp))
    return HipPosteriorSlice(h[], mu)
end

function mean_and_var(post::HipPosteriorSlice, X::AbstractMatrix{<:Real})
    Xs = Matrix{Float64}(X); M = size(Xs, 2)
    μ = Vector{Float64}(undef, M); σ2 = similar(μ); bad = Ref{Clong}(-1)
    check(ccall((:boss_gp_predict, lib), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Clong}),
        post.h, M, Xs, mean_vals(post.mean, Xs), μ, σ2, bad))
    return μ, σ2
end
mean_and_var(post::HipPosteriorSlice, x::AbstractVector{<:Real}) = first.(mean_and_var(post, hcat(x)))
mean(post::HipPosteriorSlice, x) = mean_and_var(post, x)[1]
var(post::HipPosteriorSlice, x) = mean_and_var(post, x)[2]
function BOSS.mean_and_cov(post::HipPosteriorSlice, X::AbstractMatrix{<:Real})
    Xs = Matrix{Float64}(X); M = size(Xs, 2)
    μ = Vector{Float64}(undef, M); Σ = Matrix{Float64}(undef, M, M); bad = Ref{Clong}(-1)
    check(ccall((:boss_gp_predict_cov, lib), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Clong}),
        post.h, M, Xs, mean_vals(post.mean, Xs), μ, Σ, bad))
    return μ, Σ
end
BOSS.cov(post::HipPosteriorSlice, X::AbstractMatrix{<:Real}) = BOSS.mean_and_cov(post, X)[2]

function data_loglike(m::HipGaussianProcess, data::BOSS.ExperimentData)
    X = Matrix{Float64}(data.X)
    hs = map(1:size(data.Y, 1)) do i                                   # one resident handle per output
        h = Ref{Ptr{Cvoid}}()
        check(ccall((:boss_gp_create, lib), Cint, (Cint, Cint, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{UInt8}, Ref{Ptr{Cvoid}}),
              m.device, kernel_id(m.gp.kernel), size(X, 1), size(X, 2), X, Vector{Float64}(data.Y[i, :]),
              discrete_flags(m.gp.kernel), h)); h[]
    end
    return function ll_data(p::HipGPParams)
        sum(eachindex(hs)) do i
            lp = Ref{Cdouble}()
            check(ccall((:boss_gp_update, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cdouble, Cdouble, Ptr{Cdouble}, Cint, Ref{Cdouble}),
                  hs[i], Vector{Float64}(p.λ[:, i]), p.α[i], p.σ[i], mean_vals(BOSS.mean_getindex(m.gp.mean, i), X), 0, lp))
            lp[]
        end
    end                                    # exceptions → -Inf via BOSS.safe_data_loglike, as for any model
end

# ---------------------------------------------------------------- ModelFitter (SamplingMAP semantics, batched)
Base.@kwdef struct HipBatchedMAP <: BOSS.ModelFitter{BOSS.MAPParams}
    samples::Int
end
function estimate_parameters(f::HipBatchedMAP, problem::BOSS.BossProblem, options::BOSS.BossOptions; return_all=false)
    m = problem.model::HipGaussianProcess; data = problem.data
    sampler = BOSS.params_sampler(m, data); prior = params_loglike(m)
    ps = [sampler() for _ in 1:f.samples]
    X = Matrix{Float64}(data.X); ll = zeros(f.samples)
    for i in 1:size(data.Y, 1)
        λ = reduce(hcat, (p.λ[:, i] for p in ps)); lli = zeros(f.samples); st = zeros(Cint, f.samples)
        check(ccall((:boss_gp_loglike_batch, lib), Cint,
            (Cint, Cint, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Ptr{UInt8}, Cint,
             Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cint}),
            m.device, kernel_id(m.gp.kernel), size(X, 1), size(X, 2), X, Vector{Float64}(data.Y[i, :]),
            mean_vals(BOSS.mean_getindex(m.gp.mean, i), X), 0, discrete_flags(m.gp.kernel), f.samples,
            Matrix{Float64}(λ), Float64[p.α[i] for p in ps], Float64[p.σ[i] for p in ps], lli, st))
        ll .+= lli
    end
    ll .+= prior.(ps)
    return_all && return BOSS.MAPParams.(ps, ll)
    b = argmax(ll); return BOSS.MAPParams(ps[b], ll[b])
end

# ---------------------------------------------------------------- AcquisitionMaximizer (SamplingAM semantics, batched)
Base.@kwdef struct HipBatchAM <: BOSS.AcquisitionMaximizer
    x_prior
    samples::Int
    max_attempts::Int = 200
end
function maximize_acquisition(am::HipBatchAM, problem::BOSS.BossProblem, options::BOSS.BossOptions;
                              posts = BOSS.model_posterior(problem))
    ei = problem.acquisition::BOSS.ExpectedImprovement{<:BOSS.LinFitness}
    xs = BOSS._reduce_samples([BOSS._rand_in_domain(am.x_prior, problem.domain; am.max_attempts) for _ in 1:am.samples])
    posts isa AbstractVector || (posts = [posts])
    P = BOSS.y_dim(problem); S = length(posts); M = size(xs, 2)
    hs = Ptr{Cvoid}[posts[s].slices[p].h for p in 1:P, s in 1:S]
    cand = Ref{Ptr{Cvoid}}()
    check(ccall((:boss_cand_create, lib), Cint, (Cint, Cint, Cint, Ptr{Cdouble}, Ref{Ptr{Cvoid}}),
          problem.model.device, size(xs, 1), M, Matrix{Float64}(xs), cand))
    b = BOSS.best_so_far(problem, ei.fitness)
    mask = UInt8[BOSS.in_bounds(x, problem.domain.bounds) && BOSS.in_cons(x, problem.domain.cons) for x in eachcol(xs)]
    ymax = Float64[isinf(c) ? Inf : c for c in problem.y_max]
    am_idx = Ref{Clong}(); mx = Ref{Cdouble}()
    rc = ccall((:boss_acq_ei, lib), Cint,
        (Cint, Cint, Ptr{Ptr{Cvoid}}, Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cdouble, Ptr{UInt8},
         Ptr{Cdouble}, Ref{Clong}, Ref{Cdouble}),
        P, S, hs, cand[], C_NULL #= mean_Xs: P×M×S when the GP has a prior mean =#, Float64.(ei.fitness.coefs), ymax,
        isnothing(b) ? 0 : 1, something(b, 0.0), ei.cons_safe ? mask : C_NULL, C_NULL, am_idx, mx)
    ccall((:boss_cand_free, lib), Cvoid, (Ptr{Cvoid},), cand[]); check(rc)
    return xs[:, am_idx[] + 1], mx[]
end
# ---------------------------------------------------------------- SequentialBatchAM on resident posteriors
# batch.jl:26-38 rebuilds the posterior (an O(N^3) Cholesky) for every speculative point; here the
# handles stay resident and each speculative observation is a block Cholesky append (O(N^2)).
function append!(post::HipPosteriorSlice, x::AbstractVector{<:Real}, y::Real)
    lp = Ref{Cdouble}(); X = reshape(Vector{Float64}(x), :, 1)
    check(ccall((:boss_gp_append, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Cdouble}),
          post.h, 1, X, Float64[y], mean_vals(post.mean, X), lp))
    return lp[]
end
Base.@kwdef struct HipSequentialBatchAM <: BOSS.AcquisitionMaximizer
    am::HipBatchAM
    batch_size::Int
end
function maximize_acquisition(sb::HipSequentialBatchAM, problem::BOSS.BossProblem, options::BOSS.BossOptions)
    problem_ = deepcopy(problem); post = BOSS.model_posterior(problem_)
    X = reduce(hcat, map(1:sb.batch_size) do _
        x, _ = maximize_acquisition(sb.am, problem_, options; posts = post)   # HipBatchAM with given handles
        y = BOSS.mean(post, x)
        BOSS.augment_dataset!(problem_, x, y)
        foreach(i -> append!(post.slices[i], x, y[i]), eachindex(y))
        x
    end)
    return X, nothing
end
# ---------------------------------------------------------------- analytic gradients (instead of ForwardDiff duals)
# value and gradient of the acquisition at the columns of X (d×M), one device call for all columns:
function acq_value_and_grad(problem::BOSS.BossProblem, post, X::AbstractMatrix{<:Real})
    ei = problem.acquisition::BOSS.ExpectedImprovement{<:BOSS.LinFitness}
    P = BOSS.y_dim(problem); Xs = Matrix{Float64}(X); d, M = size(Xs)
    hs = Ptr{Cvoid}[post.slices[p].h for p in 1:P]
    b = BOSS.best_so_far(problem, ei.fitness)
    mask = UInt8[BOSS.in_bounds(x, problem.domain.bounds) && BOSS.in_cons(x, problem.domain.cons) for x in eachcol(Xs)]
    acq = Vector{Float64}(undef, M); dacq = Matrix{Float64}(undef, d, M)
    check(ccall((:boss_acq_ei_grad, lib), Cint,
        (Cint, Ptr{Ptr{Cvoid}}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cdouble,
         Ptr{UInt8}, Ptr{Cdouble}, Ptr{Cdouble}),
        P, hs, M, Xs, C_NULL #= [p][M] prior means =#, C_NULL #= [p][d×M] prior-mean gradients =#,
        Float64.(ei.fitness.coefs), Float64[isinf(c) ? Inf : c for c in problem.y_max], isnothing(b) ? 0 : 1,
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
        m.device, kernel_id(m.gp.kernel), d, N, X, Vector{Float64}(data.Y[i, :]), mean_vals(mu, X), 0,
        discrete_flags(m.gp.kernel), S, λ, α, σ, ll, g, st))
    return ll, g, st          # ll[s] = -Inf and g[:, s] = 0 where st[s] != 0 (not PD / invalid parameters)
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
    return HipPosteriorSlice(h, nothing)          # mean / var / mean_and_var above apply (gradient_gp.jl:334-361)
end
function data_loglike(m::HipGradientGaussianProcess, data::BOSS.GradientData)
    h = ggp_create(m, data, 1)                    # per-output likelihood of the sliced model (gradient_gp.jl:367-397)
    return p -> try ggp_update(h, p, 1) catch e; e isa PosDefException ? -Inf : rethrow() end
end
# ---------------------------------------------------------------- NonstationaryGP (Gibbs kernel)
# The latent models stay BOSS's own (ParametrizedGP posteriors or constants); only their values cross the ABI.
struct HipNonstationaryPosterior <: BOSS.ModelPosteriorSlice{BOSS.NonstationaryGP}
    post::HipPosteriorSlice; f_λ; f_α; discrete
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
    return HipNonstationaryPosterior(HipPosteriorSlice(h[], mu), f_λ, f_α, model.discrete), lp[]   # lp = data_loglike_slice
end
function mean_and_var(p::HipNonstationaryPosterior, X::AbstractMatrix{<:Real})
    Xs = Matrix{Float64}(X); Xr = rounded(Xs, p.discrete); M = size(Xs, 2)
    μ = Vector{Float64}(undef, M); σ2 = similar(μ); bad = Ref{Clong}(-1)
    check(ccall((:boss_ngp_predict, lib), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Clong}),
        p.post.h, M, Xs, reduce(hcat, p.f_λ.(eachcol(Xr))), Float64.(p.f_α.(eachcol(Xr))), mean_vals(p.post.mean, Xs), μ, σ2, bad))
    return μ, σ2
end
end # module
