// common.hpp — shared device helpers for the bosship HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>

namespace boss {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int BLK = 128;          // block size of the blocked Cholesky / triangular solves
constexpr int PRED_RB = 256;      // rows per prediction (forward-substitution) step; handles pad N to a multiple of it
constexpr int RHS_ROWS = 128;     // extra row block carrying (y-m)^T through the factorisation
constexpr int KERN_MATERN32 = 0, KERN_MATERN52 = 1, KERN_SQEXP = 2;
constexpr int KERN_GIBBS = 3;               // internal: NonstationaryKernel (nonstationary_gp.jl:61-107), per-point λ and α
constexpr double MIN_PARAM_VALUE = 1e-8;   // src/models/gaussian_process.jl:5
constexpr double MAX_NEG_VAR = 1e-8;       // src/models/gaussian_process.jl:13
constexpr double PREDICT_JITTER = 1e-18;   // AbstractGPs default Σy of post(X*)

// v_mfma_f64_16x16x4_f64: D(16x16) = A(16x4) B(4x16) + C.
// lane l holds A[l&15][l>>4], B[l>>4][l&15]; result reg i of lane l is D[(l>>4)+4i][l&15]
// (/opt/skills/guides/cdna_hip_programming.md §3, f64 fragment layout).
__device__ __forceinline__ v4d mfma_f64(double a, double b, v4d c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ double readlane_f64(double x, int lane) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}

// 1/sqrt(p) to ~1 ulp: v_rsq_f64 seed (relative error e0 <~ 2^-23) + ONE cubically convergent
// correction  y(1 + e/2 + 3e²/8), e = 1 - p y²  →  error ~ (5/16) e0³, far below fp64 rounding.
// (This sits on the factorisation's sequential pivot chain: every dependent fp64 op costs ~12 cycles.)
__device__ __forceinline__ double rsqrt_refined(double p) {
    double y = __builtin_amdgcn_rsq(p);
    double t = p * y;
    double e = __builtin_fma(-t, y, 1.0);
    double c = __builtin_fma(0.375, e, 0.5);
    return __builtin_fma(y * e, c, y);
}

// 1/p to ~1 ulp: v_rcp_f64 seed (relative error e0 <~ 2^-23) + one cubically convergent correction
// y(1 + e + e²), e = 1 - p y  →  error ~ e0³.  Three dependent ops after the seed.
__device__ __forceinline__ double rcp_refined(double p) {
    double y = __builtin_amdgcn_rcp(p);
    double e = __builtin_fma(-p, y, 1.0);
    double t = __builtin_fma(e, e, e);
    return __builtin_fma(y, t, y);
}

// Radial profile of the base kernel on the SQUARED scaled distance.
// KernelFunctions.jl Matern32Kernel / Matern52Kernel / SqExponentialKernel.
__device__ __forceinline__ double kappa_r2(int kern, double r2) {
    if (kern == KERN_SQEXP) return exp(-0.5 * r2);
    double r = sqrt(r2);
    if (kern == KERN_MATERN32) {
        double s = 1.7320508075688772 * r;
        return (1.0 + s) * exp(-s);
    }
    double s = 2.23606797749979 * r;
    return (1.0 + s + s * s / 3.0) * exp(-s);
}

// h(r) = κ'(r)/r on the squared scaled distance (finite at r = 0): the radial factor of ∇k,
// ∇_{x*} k(x, x*) = α² h(r) (x* − x) ⊘ λ².
__device__ __forceinline__ double kappa_prime_over_r_r2(int kern, double r2) {
    if (kern == KERN_SQEXP) return -exp(-0.5 * r2);
    double r = sqrt(r2);
    if (kern == KERN_MATERN32) return -3.0 * exp(-1.7320508075688772 * r);
    double s = 2.23606797749979 * r;
    return -(5.0 / 3.0) * (1.0 + s) * exp(-s);
}

// g(r) = h'(r)/r: the second radial factor of the kernel's mixed second derivative,
// ∂²k/∂x_l∂x'_m = −α² [ h(r) δ_lm/λ_l² + g(r) s_l s_m ],  s = (x − x') ⊘ λ²   (gradient_gp.jl:143-159 obtains
// it by ForwardDiff).  Matérn-3/2 is singular at r = 0; the reference never evaluates it there (:151-152).
__device__ __forceinline__ double kappa_second_r2(int kern, double r2) {
    if (kern == KERN_SQEXP) return exp(-0.5 * r2);
    double r = sqrt(r2);
    if (kern == KERN_MATERN32) return 5.196152422706632 * exp(-1.7320508075688772 * r) / r;
    return (25.0 / 3.0) * exp(-2.23606797749979 * r);
}

// q(r) = g'(r)/r (likelihood gradient of the gradient-observation model: ∂/∂λ of the second-derivative block)
__device__ __forceinline__ double kappa_third_r2(int kern, double r2) {
    if (kern == KERN_SQEXP) return -exp(-0.5 * r2);
    double r = sqrt(r2);
    if (kern == KERN_MATERN32) return -5.196152422706632 * (1.0 + 1.7320508075688772 * r) * exp(-1.7320508075688772 * r) / (r2 * r);
    return -(25.0 * 2.23606797749979 / 3.0) * exp(-2.23606797749979 * r) / r;
}

__device__ __forceinline__ double normcdf_dev(double z) {   // StatsFuns.normcdf = erfc(-z/√2)/2
    return 0.5 * erfc(-z * 0.7071067811865476);
}
__device__ __forceinline__ double normpdf_dev(double z) {   // exp(-z²/2)/√(2π)
    return exp(-0.5 * z * z) * 0.3989422804014327;
}

}  // namespace boss
