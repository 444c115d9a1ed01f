// acq_kernels.hpp — Expected Improvement × feasibility epilogue, BI averaging, first-index arg-max, EI gradient chain
// rule, and the fp64 MFMA issue-rate probe used by bench.py.
#pragma once
#include "grad_kernels.hpp"

namespace boss {

// ------------------------------------------------------------------------------------------
// K8 EI·feas epilogue (expected_improvement.jl:68-101,113-114).  mu/var are P×M (row p at p*ldm).
// mode bit0: has best_yet, bit1: constrained.  A candidate whose variance is < -1e-8 in any
// output is poisoned with -Inf (SafeFunction semantics, src/acquisition.jl:21-25).
// The P fitness coefficients / constraints travel in the kernel arguments (no H2D copies) when
// P <= EI_MAXP, otherwise in device arrays.
// ------------------------------------------------------------------------------------------
constexpr int EI_MAXP = 16;
struct EiPar {
    int P, mode;
    double best;
    double coefs[EI_MAXP], ymax[EI_MAXP];
};

__device__ __forceinline__ double ei_value(const double* __restrict__ mu, const double* __restrict__ var, int ldm, int j,
                                           const EiPar& par, const double* __restrict__ coefs_dev,
                                           const double* __restrict__ ymax_dev) {
    const int P = par.P, mode = par.mode;
    if (mode == 0) return 0.0;                              // construct_ei(…, nothing, …, nothing): acq ≡ 0
    double muf = 0.0, vf = 0.0, fp = 1.0;
    bool poison = false;
    for (int p = 0; p < P; ++p) {
        const double cf = (P <= EI_MAXP) ? par.coefs[p] : coefs_dev[p];
        const double ym = (mode & 2) ? ((P <= EI_MAXP) ? par.ymax[p] : ymax_dev[p]) : INFINITY;
        double m = mu[(size_t)p * ldm + j], v = var[(size_t)p * ldm + j];
        if (v < 0.0) {
            if (v >= -MAX_NEG_VAR) v = 0.0;
            else poison = true;
        }
        muf = __builtin_fma(cf, m, muf);
        vf = __builtin_fma(cf * cf, v, vf);
        if ((mode & 2) && !(isinf(ym) && ym > 0.0)) {
            double s = sqrt(v);
            double z = (s == 0.0 && ym == m) ? INFINITY : (ym - m) / s;
            fp *= normcdf_dev(z);
        }
    }
    double acq;
    if (mode & 1) {
        double sf = sqrt(vf);
        double diff = muf - par.best;
        double ei;
        if (diff == 0.0 && sf == 0.0) ei = 0.0;
        else {
            double z = diff / sf;
            ei = diff * normcdf_dev(z) + sf * normpdf_dev(z);
        }
        acq = (mode & 2) ? ei * fp : ei;
    } else {
        acq = fp;
    }
    return poison ? -INFINITY : acq;
}

// the same for ONE output whose moments are in registers (rider_final_kernel)
__device__ __forceinline__ double ei_value_one(double m, double v, const EiPar& par) {
    return ei_value(&m, &v, 1, 0, par, nullptr, nullptr);
}

// BI: acq_sum[j] += acq_s(x_j) for every hyper-parameter sample but the last (the last one is folded
// into acq_epilogue_kernel).
__global__ void ei_accumulate_kernel(const double* __restrict__ mu, const double* __restrict__ var, int ldm, int M,
                                     EiPar par, const double* __restrict__ coefs_dev,
                                     const double* __restrict__ ymax_dev, double* __restrict__ acq_sum) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    acq_sum[j] += ei_value(mu, var, ldm, j, par, coefs_dev, ymax_dev);
}

// The same over samples s0 .. s1-1 whose moments lie sstride apart (the batched prediction of a set of posteriors), added in
// ascending sample order — the order of the launch-per-sample loop, so both give the same sum bit for bit.
__global__ void ei_accumulate_set_kernel(const double* __restrict__ mu, const double* __restrict__ var, size_t sstride, int ldm, int M,
                                         int s0, int s1, EiPar par, const double* __restrict__ coefs_dev,
                                         const double* __restrict__ ymax_dev, double* __restrict__ acq_sum) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    double a = acq_sum[j];
    for (int s = s0; s < s1; ++s) a += ei_value(mu + (size_t)s * sstride, var + (size_t)s * sstride, ldm, j, par, coefs_dev, ymax_dev);
    acq_sum[j] = a;
}

// K9 arg-max (Julia argmax: first index of the maximum, NaN counts as the largest value).
__device__ __forceinline__ bool better(double a, long ia, double b, long ib) {
    const bool an = a != a, bn = b != b;
    if (an != bn) return an;
    if (!an && a != b) return a > b;
    return ia < ib;
}

// Fused epilogue of one acquisition batch (one 1024-thread workgroup):
//   acq[j] = (acq_sum[j] (previous samples, if any) + acq_S(x_j)) / S, masked to 0 outside the domain
//   (make_safe, expected_improvement.jl:58-65), written back to acq_sum, and its first-index
//   arg-max, written straight into host-pinned memory (res[0] = value, res[1] = index as int64).
constexpr int ACQ_EPI_THREADS = 1024;
__global__ __launch_bounds__(ACQ_EPI_THREADS) void acq_epilogue_kernel(const double* __restrict__ mu,
                                                                       const double* __restrict__ var, int ldm, int M,
                                                                       EiPar par, const double* __restrict__ coefs_dev,
                                                                       const double* __restrict__ ymax_dev,
                                                                       double* __restrict__ acq_sum, int have_prev,
                                                                       double inv_s, const unsigned char* __restrict__ mask,
                                                                       double* __restrict__ res, unsigned long long res_seq,
                                                                       double* __restrict__ dpair = nullptr, long idx_off = 0) {
    __shared__ double sv[ACQ_EPI_THREADS];
    __shared__ long si[ACQ_EPI_THREADS];
    constexpr long NONE = 0x7fffffffffffffffL;
    double bv = -INFINITY;
    long bi = NONE;
    for (int j = threadIdx.x; j < M; j += ACQ_EPI_THREADS) {
        double a = ei_value(mu, var, ldm, j, par, coefs_dev, ymax_dev);
        if (have_prev) a += acq_sum[j];
        a *= inv_s;
        if (mask && !mask[j]) a = 0.0;
        acq_sum[j] = a;
        if (bi == NONE || better(a, j, bv, bi)) { bv = a; bi = j; }
    }
    sv[threadIdx.x] = bv;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int st = ACQ_EPI_THREADS / 2; st > 0; st >>= 1) {
        if (threadIdx.x < st) {
            const double ov = sv[threadIdx.x + st];
            const long oi = si[threadIdx.x + st];
            if (oi != NONE && (si[threadIdx.x] == NONE || better(ov, oi, sv[threadIdx.x], si[threadIdx.x]))) {
                sv[threadIdx.x] = ov;
                si[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (dpair) {                                         // one shard of a multi-device call: (max, GLOBAL index) for the all-gather
            dpair[0] = sv[0];
            dpair[1] = (double)(si[0] + idx_off);
        }
        res[0] = sv[0];
        reinterpret_cast<long*>(res)[1] = si[0];
        if (res_seq) {                                       // res is mapped host memory and the host polls this word (boss_acq_ei)
            __threadfence_system();
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(res) + 2, res_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// Acquisition value and gradient w.r.t. the candidate for one hyper-parameter sample, from the P
// outputs' moments and moment gradients (mu/var: [p][M]; dmu/dvar: [p][j*d + m]) — the chain rule
// through construct_ei (expected_improvement.jl:68-101,113-114):
//   μf = cᵀμ, σf = sqrt(c²ᵀσ²), z = (μf − b)/σf :  ∇EI = Φ(z) ∇μf + φ(z) ∇σf ,  ∇σf = c²ᵀ∇σ² / (2σf)
//   FP = Π_p Φ(t_p), t_p = (ymax_p − μ_p)/s_p, s_p = sqrt(σ²_p):
//        ∇FP = Σ_p (Π_{q≠p} Φ(t_q)) φ(t_p) ∇t_p ,  ∇t_p = −∇μ_p/s_p − (ymax_p − μ_p) ∇σ²_p / (2 s_p³)
//   acq = EI·FP (or EI, or FP, or 0 by mode); outside the domain mask: acq = 0, ∇acq = 0 (make_safe).
// Variances clipped to 0 (or exactly 0) contribute no σ-gradient.
__global__ void ei_grad_kernel(const double* __restrict__ mu, const double* __restrict__ var, const double* __restrict__ dmu,
                               const double* __restrict__ dvar, int M, int d, EiPar par,
                               const double* __restrict__ coefs_dev, const double* __restrict__ ymax_dev,
                               const unsigned char* __restrict__ mask, double* __restrict__ acq,
                               double* __restrict__ dacq) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    const int P = par.P, mode = par.mode;
    const size_t dm = (size_t)d * M;
    double* gout = dacq + (size_t)j * d;
    if (mode == 0 || (mask && !mask[j])) {
        acq[j] = 0.0;
        for (int m = 0; m < d; ++m) gout[m] = 0.0;
        return;
    }
    // pass 1: scalars
    double muf = 0.0, vf = 0.0, fp = 1.0;
    bool poison = false;
    for (int p = 0; p < P; ++p) {
        const double cf = (P <= EI_MAXP) ? par.coefs[p] : coefs_dev[p];
        const double ym = (mode & 2) ? ((P <= EI_MAXP) ? par.ymax[p] : ymax_dev[p]) : INFINITY;
        double m_ = mu[(size_t)p * M + j], v = var[(size_t)p * M + j];
        if (v < 0.0) {
            if (v >= -MAX_NEG_VAR) v = 0.0;
            else poison = true;
        }
        muf = __builtin_fma(cf, m_, muf);
        vf = __builtin_fma(cf * cf, v, vf);
        if ((mode & 2) && !(isinf(ym) && ym > 0.0)) {
            const double sd = sqrt(v);
            const double t = (sd == 0.0 && ym == m_) ? INFINITY : (ym - m_) / sd;
            fp *= normcdf_dev(t);
        }
    }
    double ei = 0.0, Phi = 0.0, phi_over_2sf = 0.0;
    if (mode & 1) {
        const double sf = sqrt(vf), diff = muf - par.best;
        if (diff == 0.0 && sf == 0.0) ei = 0.0;
        else {
            const double z = diff / sf;
            Phi = normcdf_dev(z);
            const double ph = normpdf_dev(z);
            ei = diff * Phi + sf * ph;
            phi_over_2sf = (sf > 0.0) ? ph / (2.0 * sf) : 0.0;
        }
    }
    double a;
    if (mode & 1) a = (mode & 2) ? ei * fp : ei;
    else a = fp;
    acq[j] = poison ? -INFINITY : a;
    // pass 2: gradient, one coordinate at a time
    for (int m = 0; m < d; ++m) {
        double dmuf = 0.0, dvf = 0.0, dfp = 0.0;
        for (int p = 0; p < P; ++p) {
            const double cf = (P <= EI_MAXP) ? par.coefs[p] : coefs_dev[p];
            const double gm = dmu[(size_t)p * dm + (size_t)j * d + m];
            double v = var[(size_t)p * M + j];
            const bool clipped = !(v > 0.0);
            const double gv = clipped ? 0.0 : dvar[(size_t)p * dm + (size_t)j * d + m];
            dmuf = __builtin_fma(cf, gm, dmuf);
            dvf = __builtin_fma(cf * cf, gv, dvf);
            if (mode & 2) {
                const double ym = (P <= EI_MAXP) ? par.ymax[p] : ymax_dev[p];
                if (!(isinf(ym) && ym > 0.0) && !clipped) {
                    const double m_ = mu[(size_t)p * M + j], sd = sqrt(v);
                    const double t = (ym - m_) / sd;
                    const double dt = -gm / sd - (ym - m_) * gv / (2.0 * sd * v);
                    double others = 1.0;
                    for (int q = 0; q < P; ++q) {
                        if (q == p) continue;
                        const double yq = (P <= EI_MAXP) ? par.ymax[q] : ymax_dev[q];
                        if (isinf(yq) && yq > 0.0) continue;
                        double vq = var[(size_t)q * M + j];
                        if (vq < 0.0) vq = 0.0;
                        const double mq = mu[(size_t)q * M + j], sq = sqrt(vq);
                        const double tq = (sq == 0.0 && yq == mq) ? INFINITY : (yq - mq) / sq;
                        others *= normcdf_dev(tq);
                    }
                    dfp = __builtin_fma(others * normpdf_dev(t), dt, dfp);
                }
            }
        }
        const double dei = Phi * dmuf + phi_over_2sf * dvf;
        double gA;
        if (mode & 1) gA = (mode & 2) ? dei * fp + ei * dfp : dei;
        else gA = dfp;
        gout[m] = poison ? 0.0 : gA;
    }
}

// ------------------------------------------------------------------------------------------
// issue-rate microbenchmark of v_mfma_f64_16x16x4_f64 (16 independent accumulators per wave)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mfma_f64_rate_kernel(int iters, double* __restrict__ sink) {
    v4d acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = v4d{0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] = mfma_f64(a, b, acc[t]);
    }
    double s = 0.0;
#pragma unroll
    for (int t = 0; t < 16; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    if (s == 12345.678) sink[0] = s;
}

}  // namespace boss
