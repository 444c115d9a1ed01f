// gp_kernels.hpp — all device kernels of the GP posterior / acquisition path, by theme:
//   gram_kernels.hpp     layout helpers, right-hand sides, Gram matrices (stationary, gradient observations, Gibbs)
//   predict_kernels.hpp  fused prediction, few-candidates path, resident-inverse paths, rank-one append, block inverses
//   grad_kernels.hpp     gradients w.r.t. candidates, tracked candidates, likelihood gradient
//   acq_kernels.hpp      EI × feasibility epilogue, arg-max, EI gradient, MFMA probe
#pragma once
#include "acq_kernels.hpp"
