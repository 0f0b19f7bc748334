// utils.general_loss_with_squared_residual (utils.py:264-331; Barron, "A General and Adaptive Robust Loss Function") and its
// derivative, shared by the background, warp_reg and elastic kernels.  Every branch of the reference's jnp.where ladder:
// alpha = -inf (Welsch), 0 (Cauchy), 2 (L2), +inf, else the generic power form.  (Round 2 only had the generic form: at
// alpha = 0 it reported a loss of 0 with a non-zero gradient; warp_reg_loss_alpha is a user-settable ScalarParam.)
#pragma once
#include <hip/hip_runtime.h>

namespace nrf {

// rho = scale * loss(q / scale^2, alpha);  drho = d rho / d q
__device__ __forceinline__ void general_loss_sq(float q, float alpha, float scale, float& rho, float& drho) {
  const float eps = 1.1920929e-7f;                 // jnp.finfo(float32).eps (utils.py:298)
  const float s = q / (scale * scale);             // squared_scaled_x (:301)
  const float hs = 0.5f * s;
  float loss, dloss;                               // loss(s), d loss / d s
  if (alpha == -INFINITY) {                        // :309  -expm1(-s/2)
    loss = -expm1f(-hs); dloss = 0.5f * expf(-hs);
  } else if (alpha == 0.f) {                       // :307  log1p_safe(s/2)
    const bool clamped = hs > 3e37f;
    loss = log1pf(clamped ? 3e37f : hs); dloss = clamped ? 0.f : 0.5f / (1.f + hs);
  } else if (alpha == 2.f) {                       // :305  s/2
    loss = hs; dloss = 0.5f;
  } else if (alpha == INFINITY) {                  // :311  expm1_safe(s/2)
    const bool clamped = hs > 87.5f;
    loss = expm1f(clamped ? 87.5f : hs); dloss = clamped ? 0.f : 0.5f * expf(hs);
  } else {                                         // :315-322
    const float beta = fmaxf(eps, fabsf(alpha - 2.f));
    const float a_safe = (alpha >= 0.f ? 1.f : -1.f) * fmaxf(eps, fabsf(alpha));
    const float u = s / beta + 1.f;
    loss = (beta / a_safe) * (powf(u, 0.5f * alpha) - 1.f);
    dloss = (alpha / a_safe) * 0.5f * powf(u, 0.5f * alpha - 1.f);
  }
  rho = scale * loss;
  drho = dloss / scale;
}

}  // namespace nrf
