/* bosship.h — C ABI of the MI355X-native GP-posterior + acquisition hot path for BOSS.jl.
 *
 * This is the drop-in boundary: the entry points are what a Julia `ccall`
 * layer behind BOSS's SurrogateModel / ModelFitter / AcquisitionMaximizer plugin API binds
 * (see INTEGRATION.md for the Julia stubs, and boss.jl_amd/ for the ctypes twin this pipeline
 * can execute).  Each function cites the reference code it replaces, relative to the
 * reference tree (soldasim/BOSS.jl v0.6.1).
 *
 * Conventions
 *   - all arrays are fp64, column-major, ONE OBSERVATION PER COLUMN (src/types/data.jl:10-12):
 *     X is d×N (x[k + d*j] = k-th coordinate of point j), candidates Xs are d×M.
 *   - the caller owns every host buffer; the library owns device memory behind opaque handles.
 *   - every function returns an int status (0 = BOSS_OK); no exception crosses the boundary;
 *     boss_last_error() returns a thread-local message for the last non-zero status.
 *   - the library is re-entrant across handles; one handle must not be used from two host
 *     threads at once (the reference calls its closures from Threads.@threads regions:
 *     src/utils/optim_multistart.jl:61-90, src/acquisition_maximizers/grid.jl:56-65).
 *   - there is NO CPU fallback: without a visible gfx950 device every compute entry point
 *     returns BOSS_E_NO_DEVICE.
 */
#ifndef BOSSHIP_H
#define BOSSHIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------------------ */
#define BOSS_OK            0
#define BOSS_E_INVALID     1   /* bad argument: the reference's @assert failures (gaussian_process.jl:227-233) */
#define BOSS_E_NO_DEVICE   2   /* no HIP device / HIP runtime error */
#define BOSS_E_NOT_PD      3   /* K + sigma^2 I not positive definite <-> LinearAlgebra.PosDefException */
#define BOSS_E_NEG_VAR     4   /* posterior variance < -1e-8 <-> DomainError from _clip_var (gaussian_process.jl:186-194) */
#define BOSS_E_NOT_FITTED  5   /* predict / acquisition on a handle that has no valid factorisation */
#define BOSS_E_ALLOC       6   /* device allocation failed */

/* ---- kernels: KernelFunctions.jl Matern32Kernel / Matern52Kernel (default, src/deprecated.jl:34) /
 *      SqExponentialKernel, always as alpha^2 * kernel ∘ ARDTransform(1 ./ lambda)
 *      (src/models/gaussian_process.jl:243) ------------------------------------------- */
#define BOSS_K_MATERN32 0
#define BOSS_K_MATERN52 1
#define BOSS_K_SQEXP    2

/* flags for boss_gp_update */
#define BOSS_FIT_DEFAULT   0
#define BOSS_FIT_NO_SYNC   1   /* enqueue only; logpdf_out is ignored, fetch later with boss_gp_sync */

typedef struct boss_gp   boss_gp_t;    /* one output slice's posterior: resident X, y, L, z (GaussianProcessPosterior, gaussian_process.jl:127-131) */
typedef struct boss_track boss_track_t; /* resident predictive state (V = L^-1 K*, mu, var) of one posterior at one candidate set */
typedef struct boss_cand boss_cand_t;  /* a resident batch of candidate points (the `xs` of SamplingAM.sample, sampling.jl:43-46) */

/* ---- library ---------------------------------------------------------------------- */
const char* boss_version(void);
const char* boss_last_error(void);
/* number of visible HIP devices (0 and BOSS_E_NO_DEVICE when none). */
int boss_device_count(int* n_devices_out);
/* run all work for `device` on the caller's HIP stream (e.g. torch's current stream); NULL restores the library's own stream. */
int boss_set_stream(int device, void* hip_stream);
/* block until everything enqueued on the device's stream has finished. */
int boss_device_sync(int device);

/* ---- posterior construction ---------------------------------------------------------
 * Replaces: posterior_gp / finite_gp / AbstractGPs.posterior (src/models/gaussian_process.jl:199-248)
 *           and logpdf(FiniteGP, y) (gp_data_loglike_slice, :269-280).
 *
 * boss_gp_create uploads the data of ONE output slice and keeps it resident:
 *   X        d×N, y N (row `slice` of ExperimentData.Y),
 *   discrete d flags or NULL — DiscreteKernel: flagged dims are rounded (half-to-even) in both
 *            kernel arguments (src/models/utils/kernels.jl:56-59).
 * boss_gp_update (re)builds the posterior for given hyper-parameters on the resident data:
 *   lengthscale d values, amplitude, noise_std: all must be >= 0 (else BOSS_E_INVALID); 1e-8 is
 *            ADDED to each (MIN_PARAM_VALUE, gaussian_process.jl:5,239-241);
 *   mean_X   N prior-mean values m(x_j) or NULL for the zero mean — user closures / the
 *            Semiparametric parametric mean (src/models/semiparametric.jl:79-84) are evaluated
 *            by the caller;
 *   logpdf_out  log marginal likelihood -(N log 2pi + logdet C + ||C.U'\(y-m)||^2)/2, may be NULL.
 * N is limited to 46080 observations per handle (BOSS_E_INVALID beyond).
 * On a non-PD matrix returns BOSS_E_NOT_PD, *logpdf_out = -Inf (what safe_data_loglike yields,
 * src/surrogate_model.jl:2-12) and the handle is left unfitted. */
int boss_gp_create(int device, int kernel, int d, int N, const double* X, const double* y,
                   const unsigned char* discrete, boss_gp_t** out);
int boss_gp_update(boss_gp_t* gp, const double* lengthscale, double amplitude, double noise_std,
                   const double* mean_X, int flags, double* logpdf_out);
/* wait for an update enqueued with BOSS_FIT_NO_SYNC; returns its status and logpdf. */
int boss_gp_sync(boss_gp_t* gp, double* logpdf_out);
/* one-shot convenience = create + update (the `boss_gp_fit` of SURVEY §8b). */
int boss_gp_fit(int device, int kernel, int d, int N, const double* X, const double* y,
                const double* mean_X, const double* lengthscale, double amplitude, double noise_std,
                const unsigned char* discrete, boss_gp_t** out, double* logpdf_out);
/* append-free refresh of the observations of a resident handle (same N): y only. */
int boss_gp_set_y(boss_gp_t* gp, const double* y);
/* Block Cholesky append.  Replaces: augment_dataset!(problem, x, y)
 * (src/types/problem.jl:191-198) followed by the model_posterior(problem) that
 * SequentialBatchAM's speculative loop recomputes from scratch for every selected point
 * (src/acquisition_maximizers/batch.jl:26-38), i.e. posterior_gp (gaussian_process.jl:199-211)
 * on the augmented data with the SAME hyper-parameters as the last boss_gp_update.
 * X_new is d×n column-major, y_new has n entries, mean_new is the prior mean at the new points
 * (n entries, or NULL for zero).  Only the 128-row blocks that contain new observations are
 * rebuilt (O(N^2) work per block instead of the O(N^3) re-factorisation); the result — factor,
 * z = L^{-1}(y-m), logpdf of all N+n observations — equals a fresh fit of the augmented data
 * to rounding.  The handle must be fitted; device storage grows as needed.  A posterior that was updated with a
 * prior mean (mean_X) requires mean_new (BOSS_E_INVALID otherwise).
 * From the second single-observation append after an update (the pattern of SequentialBatchAM and of
 * a BO loop that keeps its hyper-parameters) the handle keeps both inverse factors resident and an
 * append is one pass over each of them (rank-one append: L <- [L 0; l' d], L^-1 <- [L^-1 0; -w'/d 1/d]);
 * BOSS_WINV_AFTER=0 in the environment keeps every append on the block path. */
int boss_gp_append(boss_gp_t* gp, int n, const double* X_new, const double* y_new, const double* mean_new,
                   double* logpdf_out);
/* number of observations currently in the handle (after a boss_gp_append that failed part-way — n in 2..8 single
 * appends on resident inverse factors — it tells how many of them were taken). */
int boss_gp_n(const boss_gp_t* gp, int* n_out);
/* Reserve device storage for N_total observations (appends up to that size then need no
 * re-allocation).  Leaves the handle unfitted: follow with boss_gp_update. */
int boss_gp_reserve(boss_gp_t* gp, int N_total);
void boss_gp_free(boss_gp_t* gp);

/* Gradient of the log marginal likelihood w.r.t. the hyper-parameters.  Replaces: the
 * derivative of `loglike` (data_loglike, gaussian_process.jl:250-280) that OptimizationMAP's
 * optimiser obtains by automatic differentiation (src/model_fitters/optimization.jl:146-164).
 * Evaluated at the hyper-parameters of the last boss_gp_update on this handle:
 *   grad_out[0..d-1] = d logpdf / d lengthscale_m,  grad_out[d] = d/d amplitude,  grad_out[d+1] = d/d noise_std
 *   (d logpdf/d theta = 1/2 tr((a a' - K^-1) dK/dtheta), K^-1 = L^-T L^-1 formed on the device);
 *   logpdf_out (may be NULL) returns the value again.  The prior mean is treated as constant in theta.
 * Costs about one acquisition pass (triangular inverse) + a K = N syrk. */
int boss_gp_loglike_grad(boss_gp_t* gp, double* logpdf_out, double* grad_out);

/* introspection for parity tests: lower Cholesky factor L (N×N column-major, upper part zeroed)
 * and z = L \ (y - m) (N). Either pointer may be NULL. */
int boss_gp_get_factor(const boss_gp_t* gp, double* L_out, double* z_out);

/* ---- gradient observations ----------------------------------------------------
 * Replaces: GradientGaussianProcess — model_posterior_slice (src/models/gradient_gp.jl:307-329),
 *           data_loglike (:367-397), mean / var / mean_and_var (:334-361).
 *
 * Every point carries its value and its gradient: n points give an n(1+d) system over the
 * observation ordering [y_1..n, dy/dx_1 (1..n), ..., dy/dx_d (1..n)] (_build_obs_vector, :288-302).
 * boss_ggp_create keeps one output slice resident:
 *   X   d×n,  y  n,  dY  d×n column-major (dY[l + d*j] = dy/dx_l at x_j — `data.dY[slice, :, :]`).
 * boss_ggp_update builds the augmented Gram matrix (_build_augmented_kernel, :175-210: value, first and
 * mixed second derivatives of the kernel, evaluated at (x_i, x_j + 1e-8) for coincident points; noise
 * (sigma+1e-8)^2 on the value block and (grad_noise_std+1e-8)^2 on the gradient blocks; the upper
 * triangle is the one that counts, `Symmetric(K)`), factorises it and returns
 *   logpdf_out = -(y~' K^-1 y~ + log|K| + n(1+d) log 2pi)/2.
 * The model's mean function is not used by the reference for this model and is not taken here.
 * The returned handle is a boss_gp_t: boss_gp_sync, boss_gp_predict (mu = k*'alpha, var = max(0, k(x,x) - |L^-1 k*|^2),
 * k* from _build_cross_cov :221-243; mean_Xs must be NULL), boss_gp_get_factor, boss_acq_ei, boss_acq_ei_moments
 * boss_gp_predict_grad and boss_acq_ei_grad (the gradients ForwardDiff pushes through :334-361 inside OptimizationAM,
 * src/acquisition_maximizers/optimization.jl:36,89-118; mean_Xs / mean_grad must be NULL; var is max(0, .), its gradient that of
 * the unclipped expression) and boss_gp_free work on it; boss_ggp_loglike_grad and boss_ggp_append are its own forms of the
 * likelihood gradient and of augment_dataset!; the entry points that assume value-only observations (boss_gp_update,
 * boss_gp_set_y, boss_gp_append, boss_gp_reserve, boss_gp_predict_cov, boss_gp_loglike_grad, boss_track_create) return
 * BOSS_E_INVALID.
 * Limits: d <= 16, n(1+d) <= 46080. */
int boss_ggp_create(int device, int kernel, int d, int n, const double* X, const double* y, const double* dY,
                    boss_gp_t** out);
int boss_ggp_update(boss_gp_t* gp, const double* lengthscale, double amplitude, double noise_std,
                    double grad_noise_std, int flags, double* logpdf_out);
/* data_loglike of the gradient-observation model with its gradient, as OptimizationMAP takes it through ForwardDiff
 * (src/model_fitters/optimization.jl:146-164 through gradient_gp.jl:367-397): at the parameters of the last boss_ggp_update,
 *   grad_out[d + 3] = d logpdf / d(lengthscale[0..d-1], amplitude, noise_std, grad_noise_std)
 * (= 1/2 sum_ab (a a' - K^-1)_ab dK_ab/dtheta over the matrix `cholesky(Symmetric(K))` sees, a = K^-1 y~; the lengthscale derivatives
 * of the second-derivative block need the third radial profile of the kernel).  logpdf_out may be NULL. */
int boss_ggp_loglike_grad(boss_gp_t* gp, double* logpdf_out, double* grad_out);
/* augment_dataset! (src/types/problem.jl:191-198) + the posterior at unchanged hyper-parameters: n_new further points with values
 * and gradients (X_new d×n_new, y_new n_new, dY_new d×n_new column-major).  New observations land inside every block of the
 * ordering [y; dy/dx_1; ...; dy/dx_d], so — as in the reference — the augmented system is rebuilt and factorised again; the handle
 * stays the same object.  logpdf_out: logpdf of all n + n_new points.  Needs a fitted handle (BOSS_E_NOT_FITTED otherwise). */
int boss_ggp_append(boss_gp_t* gp, int n_new, const double* X_new, const double* y_new, const double* dY_new, double* logpdf_out);

/* ---- nonstationary posteriors -----------------------------------------------------
 * Replaces: NonstationaryGP — NonstationaryKernel / gibbs_kernel (src/models/nonstationary_gp/nonstationary_gp.jl:61-107),
 *           finite_nongp (:153-196), model_posterior_slice (:153-157), data_loglike_slice (:237-245).
 *
 *   k(x, y) = ((a(x) + a(y))/2)^2 prod_i sqrt(2 l_i(x) l_i(y) / (l_i(x)^2 + l_i(y)^2)) exp(-(x_i - y_i)^2 / (l_i(x)^2 + l_i(y)^2)),
 *   noise s(x_j)^2 on the diagonal.  l(.), a(.), s(.) are the caller's latent models (ParametrizedGP posteriors or
 *   constants, _param_posterior_slice :198-212) evaluated on the host; they cross the ABI as arrays:
 *   lam_X d×N (column j = l(x_j)), amp_X N, noise_X N;  lam_Xs d×M, amp_Xs M at the candidates.
 *   Dims flagged discrete are rounded in both kernel arguments (make_discrete, :183-191); the latent models must
 *   then be evaluated at the rounded points as well.  No 1e-8 is added to these parameters (the reference adds none).
 * boss_ngp_update factorises and returns logpdf(FiniteGP, y); boss_ngp_predict is mean_and_var with _clip_var
 * (k(x*,x*) = a(x*)^2, +1e-18 jitter as for the plain model).  boss_gp_sync, boss_gp_set_y, boss_gp_get_factor,
 * boss_gp_free, boss_acq_ei_moments (EI on the predicted moments), boss_ngp_predict_grad and boss_acq_ei_grad_moments (their
 * gradients w.r.t. the candidates) work with these handles; the other
 * boss_gp_* / boss_acq_* / boss_track_* entry points return BOSS_E_INVALID for them. */
int boss_ngp_create(int device, int d, int N, const double* X, const double* y, const unsigned char* discrete,
                    boss_gp_t** out);
int boss_ngp_update(boss_gp_t* gp, const double* lam_X, const double* amp_X, const double* noise_X,
                    const double* mean_X, int flags, double* logpdf_out);
/* augment_dataset! (src/types/problem.jl:191-198) for a fitted nonstationary posterior: n_new further observations with the latent
 * models' values AT THE NEW POINTS (lam_new d×n_new, amp_new, noise_new; mean_new when the posterior has a prior mean); the values
 * at the old points are the resident ones.  Every kernel entry couples both points' lengthscales, so — as in the reference — the
 * system is rebuilt and factorised; the handle stays the same object.  Needs a fitted handle (BOSS_E_NOT_FITTED otherwise). */
int boss_ngp_append(boss_gp_t* gp, int n_new, const double* X_new, const double* y_new, const double* lam_new,
                    const double* amp_new, const double* noise_new, const double* mean_new, double* logpdf_out);
/* data_loglike_slice (nonstationary_gp.jl:237-245) with its partial derivatives w.r.t. the latent models' VALUES at the training
 * points — what a gradient-based fitter's AD carries back to the latent models through finite_nongp (:183-196); the caller chains
 * them through its own latent models.  At the parameters of the last boss_ngp_update; every output may be NULL:
 *   dlam_out d×N (column j = d logpdf / d l(x_j), the layout of lam_X), damp_out N, dnoise_out N, dmean_out N (= K^-1 (y - m)).
 * x_dim <= 16. */
int boss_ngp_loglike_grad(boss_gp_t* gp, double* logpdf_out, double* dlam_out, double* damp_out, double* dnoise_out,
                          double* dmean_out);
int boss_ngp_predict(boss_gp_t* gp, int M, const double* Xs, const double* lam_Xs, const double* amp_Xs,
                     const double* mean_Xs, double* mu, double* var, long* bad_index);
/* mean_and_var of a nonstationary posterior AND its gradient w.r.t. the candidates.
 * Replaces: the derivatives ForwardDiff pushes through nonstationary_gp.jl:153-196 inside OptimizationAM
 * (src/acquisition_maximizers/optimization.jl:36,89-118).  The candidate enters the Gibbs kernel directly and through the
 * latent l(x*), a(x*), whose Jacobians arrive evaluated like their values (the latent models are the caller's):
 *   dlam_Xs d×d×M, dlam_Xs[l + d*(m + d*j)] = d l_l / d x_m at candidate j, or NULL (constant lengthscales);
 *   damp_Xs d×M,   damp_Xs[m + d*j]         = d a / d x_m,                  or NULL (constant amplitude);
 *   mean_Xs M / mean_grad d×M prior mean and its gradient at the candidates, or NULL;
 *   mu, var (clipped like boss_ngp_predict), dmu, dvar d×M column-major (gradient of the UNclipped variance).
 * Dims flagged discrete get a zero explicit part; the caller passes zero Jacobian columns for them.  x_dim <= 16. */
int boss_ngp_predict_grad(boss_gp_t* gp, int M, const double* Xs, const double* lam_Xs, const double* amp_Xs,
                          const double* dlam_Xs, const double* damp_Xs, const double* mean_Xs, const double* mean_grad,
                          double* mu, double* var, double* dmu, double* dvar, long* bad_index);

/* ---- batched log-likelihood -----------------------------------------------------------
 * Replaces: the `loglike.(samples)` loop of SamplingMAP (src/model_fitters/sampling.jl:59-78) and the
 * per-sample likelihood calls of OptimizationMAP / TuringBI (src/model_fitters/optimization.jl:153-160,
 * ext/TuringExt.jl:78-86) for S hyper-parameter sets on the same (X, y) slice.
 *   lengthscales d×S (column s = set s), amplitudes S, noise_stds S,
 *   mean_X NULL, or N values shared by all sets (mean_stride = 0), or S×N with mean_stride = N.
 *   ll_out S (‑Inf where not PD), status_out S (BOSS_OK / BOSS_E_NOT_PD / BOSS_E_INVALID) or NULL. */
int boss_gp_loglike_batch(int device, int kernel, int d, int N, const double* X, const double* y,
                          const double* mean_X, int mean_stride, const unsigned char* discrete,
                          int S, const double* lengthscales, const double* amplitudes,
                          const double* noise_stds, double* ll_out, int* status_out);
/* The same with the gradient of every log-likelihood w.r.t. (lengthscale[d], amplitude, noise_std):
 * grad_out is (d+2)×S (column s = set s; zeros where the status is not BOSS_OK).  Replaces the S value-and-gradient
 * evaluations one round of a multistart OptimizationMAP makes (src/model_fitters/optimization.jl:146-164, gradients by
 * automatic differentiation there).  The factorisations run batched; d <= 32. */
int boss_gp_loglike_grad_batch(int device, int kernel, int d, int N, const double* X, const double* y,
                               const double* mean_X, int mean_stride, const unsigned char* discrete,
                               int S, const double* lengthscales, const double* amplitudes,
                               const double* noise_stds, double* ll_out, double* grad_out, int* status_out);
/* S RESIDENT posteriors of one output slice out of ONE batched factorisation.
 * Replaces: the broadcast `model_posterior.(Ref(model), params, Ref(data))` over the S parameter samples of a Bayesian-inference
 * fit (src/posterior.jl:15-19; samples from ext/TuringExt.jl:88-107), i.e. S calls of posterior_gp (gaussian_process.jl:199-211)
 * on the same (X, y) slice.  Arguments as boss_gp_loglike_batch; X and y are uploaded once and shared by the S members.
 *   out S handles — ordinary boss_gp_t (predict, acquisition, gradients, update, append, free): each is freed with boss_gp_free;
 *       the shared device storage goes with the last one.  A member whose status is not BOSS_OK is returned unfitted.
 *   logpdf_out S log marginal likelihoods (-Inf where not PD) or NULL, status_out S (BOSS_OK / BOSS_E_NOT_PD / BOSS_E_INVALID) or NULL.
 * boss_acq_ei walks S > 1 equally shaped posteriors of an output in one prediction launch (grid = candidate tiles × samples). */
int boss_gp_fit_batch(int device, int kernel, int d, int N, const double* X, const double* y,
                      const double* mean_X, int mean_stride, const unsigned char* discrete,
                      int S, const double* lengthscales, const double* amplitudes,
                      const double* noise_stds, boss_gp_t** out, double* logpdf_out, int* status_out);

/* ---- prediction -------------------------------------------------------------------------
 * Replaces: mean_and_var(post, X::Matrix) (gaussian_process.jl:174-178) =
 *   mu = m(X*) + K*' a ; V = C.U' \ K* ; var = k(x*,x*) - sum_i V_ij^2 + 1e-18 ; _clip_var.
 *   Xs d×M, mean_Xs M values m(x*_j) or NULL; mu, var: M each.
 * Returns BOSS_E_NEG_VAR and the first offending index in *bad_index (else -1) when a variance is
 * below -1e-8 (DomainError); mu/var are still written (var unclipped at the offending entries).
 * Calls with one to four candidates — the reference evaluates the acquisition one point at a time
 * (expected_improvement.jl:75,79) — are served from an explicit inverse factor that the handle builds
 * on the second such call after an update / append (≈1 ms once, N^2 doubles of extra device memory;
 * BOSS_WINV_AFTER=0 in the environment disables it). */
int boss_gp_predict(boss_gp_t* gp, int M, const double* Xs, const double* mean_Xs,
                    double* mu, double* var, long* bad_index);

/* Posterior moments AND their analytic gradients w.r.t. the candidate coordinates.
 * Replaces: the derivative of mean_and_var(post, x) that the reference obtains by pushing
 * ForwardDiff duals through AbstractGPs when OptimizationAM refines candidates
 * (src/acquisition_maximizers/optimization.jl:36 AutoForwardDiff, :89-118; acquisition closure
 * src/acquisitions/expected_improvement.jl:74-84):
 *   dmu[:,j]  = grad m(x*_j) + sum_i a_i grad k(x_i, x*_j),   a = (K+sigma^2 I)^-1 (y - m)
 *   dvar[:,j] = -2 sum_i w_ij grad k(x_i, x*_j),              w_j = (K+sigma^2 I)^-1 k*_j
 *   Xs d×M; mean_Xs M or NULL; mean_grad d×M (gradient of the prior mean at the candidates) or NULL;
 *   mu, var M each (var through _clip_var, as boss_gp_predict); dmu, dvar d×M column-major.
 * Dimensions flagged discrete are rounded inside the kernel (DiscreteKernel), their gradient is 0.
 * Costs about two boss_gp_predict passes (forward substitution + its adjoint). */
int boss_gp_predict_grad(boss_gp_t* gp, int M, const double* Xs, const double* mean_Xs, const double* mean_grad,
                         double* mu, double* var, double* dmu, double* dvar, long* bad_index);

/* Replaces: mean_and_cov(post, X::Matrix) / cov (gaussian_process.jl:163-167,180-184):
 *   Sigma = K** - V'V + 1e-18 I (M×M, column-major, full symmetric), diagonal through _clip_var.
 * Not on the acquisition path (EI only needs the diagonal); provided so the whole posterior API
 * of the plugin is served from the device-resident factor. */
int boss_gp_predict_cov(boss_gp_t* gp, int M, const double* Xs, const double* mean_Xs,
                        double* mu, double* cov, long* bad_index);

/* resident candidates (one upload, many acquisition passes / many posteriors) */
int boss_cand_create(int device, int d, int M, const double* Xs, boss_cand_t** out);
void boss_cand_free(boss_cand_t* cand);

/* ---- acquisition ------------------------------------------------------------------------
 * Replaces: the `vals = acq.(eachcol(xs)); argmax(vals)` loop of SamplingAM / GridAM
 * (src/acquisition_maximizers/sampling.jl:43-46,36-39; grid.jl:52-65) with acq built by
 * construct_acquisition(::ExpectedImprovement) (src/acquisitions/expected_improvement.jl:49-90):
 *   gps       P×S handles, gps[p + P*s] = output p of hyper-parameter sample s (S = 1 for MAP);
 *             the acquisition is averaged over s (:87-90);
 *   mean_Xs   NULL or P×M×S prior means at the candidates, index p + P*(j + M*s);
 *   fit_coefs P LinFitness coefficients (src/types/fitness.jl);
 *   y_max     P upper constraints, +Inf allowed (-> factor 1, src/utils/inf.jl), or NULL for
 *             `constraints === nothing`;
 *   has_best/best  best_so_far(...) (:134-140) — has_best = 0 means `nothing`;
 *   valid_mask M bytes or NULL: 0 -> acq = 0.0 (make_safe, :58-65: outside bounds / cons, evaluated
 *             by the caller because `cons` is a user closure);
 *   acq_out   M values or NULL; a candidate whose variance raises DomainError gets -Inf, as the
 *             reference's SafeFunction wrapper does (src/acquisition.jl:21-25);
 *   argmax_out / max_out  first index of the maximum (Julia argmax) over this batch and its value.
 * All handles and `cand` must live on the same device. */
int boss_acq_ei(int P, int S, boss_gp_t* const* gps, const boss_cand_t* cand,
                const double* mean_Xs, const double* fit_coefs, const double* y_max,
                int has_best, double best, const unsigned char* valid_mask,
                double* acq_out, long* argmax_out, double* max_out);

/* One BO iteration's model update AND its first acquisition in one call: the candidates' forward substitution rides along
 * the factorisation (csrc/rider.hpp) instead of starting when it has ended.
 * Replaces: the pair `estimate_parameters!` -> `maximize_acquisition` of src/bo.jl:30-48 for maximisers whose candidates
 * do not depend on the posterior (SamplingAM, src/acquisition_maximizers/sampling.jl:43-57; GridAM, grid.jl:52-65) —
 * i.e. boss_gp_update (posterior_gp, src/models/gaussian_process.jl:199-211) followed by boss_acq_ei (P = 1, S = 1:
 * mean_and_var :174-178 + construct_ei, src/acquisitions/expected_improvement.jl:68-101) on the SAME handle.
 *   gp, lengthscale, amplitude, noise_std, mean_X: as boss_gp_update;   cand: resident candidates on gp's device;
 *   mean_Xs  NULL or M prior means at the candidates;   fit_coef: the LinFitness coefficient of this output;
 *   y_max    the output's upper constraint (+Inf allowed: factor 1, src/utils/inf.jl); NaN = `constraints === nothing`
 *            (boss_acq_ei's y_max == NULL);   has_best / best, valid_mask: as boss_acq_ei;
 *   logpdf_out  the update's log marginal likelihood;   mu_out / var_out  M posterior moments (unclipped: what
 *       mean_and_var returns before _clip_var; the EI epilogue clips) or NULL;   acq_out  M values or NULL;
 *   argmax_out / max_out as boss_acq_ei;   fused_out (or NULL): 1 = the substitution rode along, 0 = the call ran the two
 *       phases one after the other (no resident chain for this size / device / stream, N <= 128, x_dim > 32, a fallback).
 * The numbers do not depend on which way the call went beyond the stated fp64 tolerance (the order of the floating-point
 * sums differs); repeated calls on one path are bit-identical.  Errors: as boss_gp_update (BOSS_E_NOT_PD leaves the handle
 * unfitted and the acquisition outputs untouched). */
int boss_gp_update_acq(boss_gp_t* gp, const double* lengthscale, double amplitude, double noise_std, const double* mean_X,
                       const boss_cand_t* cand, const double* mean_Xs, double fit_coef, double y_max, int has_best,
                       double best, const unsigned char* valid_mask, double* logpdf_out, double* mu_out, double* var_out,
                       double* acq_out, long* argmax_out, double* max_out, int* fused_out);

/* The same EI x feasibility epilogue (expected_improvement.jl:68-101,113-114) and arg-max, but
 * from posterior moments the caller already holds: mu/var are S×P×M, index j + M*(p + P*s)
 * (row p of sample s is what mean_and_var(post_s, Xs)[p, :] returns, src/posterior.jl:60-72).
 * Used when the P outputs are fitted on different GPUs and their (mu, var) rows were all-gathered
 * (one process per GPU; see boss.jl_amd/distributed.py).  var < -1e-8 poisons the candidate
 * with -Inf as in boss_acq_ei. */
int boss_acq_ei_moments(int device, int P, int S, int M, const double* mu, const double* var,
                        const double* fit_coefs, const double* y_max, int has_best, double best,
                        const unsigned char* valid_mask, double* acq_out, long* argmax_out,
                        double* max_out);

/* The chain rule of boss_acq_ei_grad (construct_ei, expected_improvement.jl:68-101,113-114) from moments and moment gradients the
 * caller already holds — boss_ngp_predict_grad per output of a nonstationary model, or outputs fitted on other ranks:
 *   mu / var P×M (row p at p*M; var clipped), dmu / dvar P blocks of d×M column-major;  the other arguments as boss_acq_ei_grad. */
int boss_acq_ei_grad_moments(int device, int P, int M, int d, const double* mu, const double* var, const double* dmu,
                             const double* dvar, const double* fit_coefs, const double* y_max, int has_best, double best,
                             const unsigned char* valid_mask, double* acq_out, double* dacq_out);

/* Acquisition value AND its gradient w.r.t. the candidates, for one hyper-parameter
 * sample (MAP).  Replaces: differentiating construct_acquisition(::ExpectedImprovement)
 * (src/acquisitions/expected_improvement.jl:49-101,113-114) with ForwardDiff inside OptimizationAM's
 * multistart refinement (src/acquisition_maximizers/optimization.jl:36,89-118):
 *   gps        P handles (one per output), all on one device;
 *   Xs         d×M candidates (host);  mean_Xs NULL or [p][M];  mean_grad NULL or [p][d×M];
 *   fit_coefs, y_max, has_best/best, valid_mask: as boss_acq_ei;
 *   acq_out    M values;  dacq_out d×M column-major (gradient of acq_out[j] w.r.t. candidate j).
 * The EI x feasibility chain rule runs on the device behind boss_gp_predict_grad's moment gradients. */
int boss_acq_ei_grad(int P, boss_gp_t* const* gps, int M, const double* Xs, const double* mean_Xs,
                     const double* mean_grad, const double* fit_coefs, const double* y_max, int has_best,
                     double best, const unsigned char* valid_mask, double* acq_out, double* dacq_out);

/* ---- tracked candidates -----------------------------------------------------------------------
 * SequentialBatchAM (src/acquisition_maximizers/batch.jl:26-38) re-evaluates the acquisition on
 * the whole candidate set after every speculative observation; with a FIXED candidate set (GridAM
 * points, or a seeded sample) the forward-substitution result V = C.U' \ K* can stay resident and
 * be EXTENDED by one row per appended observation (boss_gp_append) — an O(N M) update of mu and
 * var instead of the O(N^2 M) re-solve:
 *   boss_track_create   runs the prediction once and keeps V, mu, var (unclipped) on the device;
 *                       mean_Xs: prior mean at the candidates (M values) or NULL;
 *   boss_track_sync     brings the state up to the posterior's current N (no-op when in sync);
 *   boss_track_moments  copies mu / var (unclipped, like mean_and_var before _clip_var) of the
 *                       candidates [first, first+count) to the host (syncs first);
 *   boss_acq_ei_tracks  = boss_acq_ei on tracked states (tracks[p + P*s]; syncs them first).
 * A track is bound to the hyper-parameters of the boss_gp_update that preceded its creation:
 * after another boss_gp_update every call returns BOSS_E_INVALID (create a new track).  The
 * posterior handle must outlive its tracks. */
int boss_track_create(boss_gp_t* gp, const boss_cand_t* cand, const double* mean_Xs, boss_track_t** out);
void boss_track_free(boss_track_t* track);
int boss_track_sync(boss_track_t* track);
int boss_track_moments(boss_track_t* track, int first, int count, double* mu, double* var);
int boss_acq_ei_tracks(int P, int S, boss_track_t* const* tracks, const double* fit_coefs, const double* y_max,
                       int has_best, double best, const unsigned char* valid_mask,
                       double* acq_out, long* argmax_out, double* max_out);

/* ---- several GPUs from ONE process (north_star: "shard embarrassingly across the 8 GPUs of one node with a
 * single RCCL allreduce over xGMI for the argmax") ---------------------------------------------------------------
 * What a Julia caller — no torch, no MPI — uses to spread the acquisition path over the GPUs of a node.  The devices
 * work concurrently (one host thread per device inside a call); the exchanges run over RCCL, which the library loads
 * at run time (dlopen librccl.so.1, ncclCommInitAll over all devices).  With one device and no RCCL the calls still work
 * (the exchange is the identity); with several devices and no RCCL boss_init fails (BOSS_E_NO_DEVICE).
 *
 * boss_init          enumerate the visible devices (BOSS_MAX_DEVICES in the environment caps the count), open a context
 *                    and a communicator on each.  Idempotent.  Replaces nothing in the reference (its parallelism is
 *                    Threads.@threads on the host: src/utils/optim_multistart.jl:61-90).
 * boss_comm_info     number of devices opened by boss_init and whether the exchanges run over RCCL.
 * boss_shutdown      destroy the communicators and exchange buffers (posterior handles are the caller's to free first). */
int  boss_init(int* n_devices_out);
int  boss_comm_info(int* n_devices_out, int* rccl_out);
void boss_shutdown(void);
/* The same hyper-parameters on G replicas of one posterior (gps[g] created by boss_gp_create on device g): every
 * device factorises concurrently; *logpdf_out from replica 0 (the factorisation is deterministic: all replicas agree). */
int boss_multi_gp_update(int G, boss_gp_t* const* gps, const double* lengthscale, double amplitude, double noise_std,
                         const double* mean_X, double* logpdf_out);
/* Candidates sharded over G devices (BASELINE config 3).  Replaces the same loop as boss_acq_ei
 * (src/acquisition_maximizers/sampling.jl:43-57) with the candidate columns split M/G:
 *   gps      G×S×P handles, gps[p + P*(s + S*g)] = replica on device g of output p, sample s;
 *   Xs       d×M candidates on the host; device g evaluates the contiguous, balanced shard g of the columns;
 *   mean_Xs, fit_coefs, y_max, has_best/best, valid_mask, acq_out (M values or NULL): as boss_acq_ei, for all M;
 *   argmax_out / max_out: the GLOBAL first-index arg-max — every device contributes its (max, global index) pair to ONE
 *   RCCL all-gather (16 bytes per rank; RCCL has no MAXLOC), reduced with Julia's argmax rules (ties: smaller index,
 *   NaN largest).  G must equal boss_init's count for the exchange to run over RCCL (a smaller G reduces on the host). */
int boss_multi_acq_ei(int G, int P, int S, boss_gp_t* const* gps, int M, const double* Xs, const double* mean_Xs,
                      const double* fit_coefs, const double* y_max, int has_best, double best,
                      const unsigned char* valid_mask, double* acq_out, long* argmax_out, double* max_out);
/* Outputs sharded (BASELINE config 4: "outputs sharded one-per-GPU"): gps[p + P*s] may live on ANY device opened by
 * boss_init (model_posterior_slice per output, src/models/gaussian_process.jl:133-141, fitted where it lives).  Each
 * owner predicts its (mu, var) rows for all M candidates; the rows travel in ONE RCCL all-reduce(sum) over zero-filled
 * blocks (exact), then the EI x feasibility epilogue and the arg-max run on the device of gps[0]
 * (expected_improvement.jl:68-90).  Arguments as boss_acq_ei with host candidates. */
// Resident candidate shards for boss_multi_acq_ei_cand: shard g (the balanced contiguous range of get_sample_counts,
// /root/reference/src/utils/sampling.jl:6-13) is uploaded to device g once; replaces the per-call `acq.(eachcol(xs))` operand of
// /root/reference/src/acquisition_maximizers/sampling.jl:43-57 when several acquisition calls share their candidates.
typedef struct boss_multi_cand boss_multi_cand_t;
int boss_multi_cand_create(int G, int d, int M, const double* Xs /*d×M*/, boss_multi_cand_t** out);
void boss_multi_cand_free(boss_multi_cand_t* cand);
int boss_multi_acq_ei_cand(int G, int P, int S, boss_gp_t* const* gps /*[G][S][P]*/, const boss_multi_cand_t* cand,
                           const double* mean_Xs /*P×M×S|NULL*/, const double* fit_coefs, const double* y_max, int has_best, double best,
                           const unsigned char* valid_mask, double* acq_out, long* argmax_out, double* max_out);
int boss_multi_acq_ei_outputs(int P, int S, boss_gp_t* const* gps, int M, const double* Xs, const double* mean_Xs,
                              const double* fit_coefs, const double* y_max, int has_best, double best,
                              const unsigned char* valid_mask, double* acq_out, long* argmax_out, double* max_out);
/* Hyper-parameter samples sharded (BASELINE config 5): all P outputs of sample s live on one device, different samples
 * on different devices.  Every device sums acq_s(x_j) over ITS samples; ONE RCCL all-reduce(sum) of M doubles; the mean
 * over S (src/acquisitions/expected_improvement.jl:87-90), make_safe mask and arg-max on the device of sample 0. */
int boss_multi_acq_ei_samples(int P, int S, boss_gp_t* const* gps, int M, const double* Xs, const double* mean_Xs,
                              const double* fit_coefs, const double* y_max, int has_best, double best,
                              const unsigned char* valid_mask, double* acq_out, long* argmax_out, double* max_out);
/* boss_gp_loglike_batch with the S hyper-parameter sets split over devices 0..G-1 (contiguous, balanced); no collective:
 * every device writes its slice of ll_out / status_out. */
int boss_multi_loglike_batch(int G, int kernel, int d, int N, const double* X, const double* y, const double* mean_X,
                             int mean_stride, const unsigned char* discrete, int S, const double* lengthscales,
                             const double* amplitudes, const double* noise_stds, double* ll_out, int* status_out);

/* ---- measurement helpers (bench.py / profiles) ------------------------------------------ */
/* issue-rate microbenchmark of v_mfma_f64_16x16x4_f64: every SIMD of the device issues
 * `iters` x 16 independent MFMAs; returns the achieved TFLOP/s (calibrates the fp64 MFMA peak
 * that /opt/skills/guides/MI355X_MICROARCH.md does not list). */
int boss_bench_mfma_f64(int device, int iters, double* tflops_out);
/* per-kernel-class HIP-event timing: enable, run work, then read back (ms_total, launches). */
int boss_prof_enable(int device, int on);
int boss_prof_reset(int device);
int boss_prof_get(int device, const char* kernel_class, double* ms_total, long* launches);

#ifdef __cplusplus
}
#endif
#endif /* BOSSHIP_H */
