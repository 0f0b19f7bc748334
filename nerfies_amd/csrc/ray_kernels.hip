// Per-ray kernels of the nerfies hot path for gfx950 (one 64-lane wave per ray; prefix
// products / sums and the inverse-CDF search done with wave shuffles and wave-private LDS).
//
// Replaces (reference, /root/reference/nerfies):
//   model_utils.sample_along_rays        model_utils.py:36-73
//   model_utils.volumetric_rendering     model_utils.py:76-136 (+ depth helpers :218-263)
//   model_utils.piecewise_constant_pdf / sample_pdf   model_utils.py:139-215
//   models.NerfModel.get_condition_inputs (viewdir posenc + GLO gathers)  models.py:186-228
//   training._compute_loss_and_stats MSE/psnr   training.py:172,225 ; utils.py:94-103
//   flax.optim.Adam.apply_gradient        training.py:268-269
#include "nrf_internal.h"
#include "general_loss.h"
#include "philox.h"

namespace nrf {

// ------------------------------------------------------------------ wave helpers (64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_incl_sum(float v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const float t = __shfl_up(v, o); if (lane >= o) v += t; }
  return v;
}
__device__ __forceinline__ float wave_incl_prod(float v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const float t = __shfl_up(v, o); if (lane >= o) v *= t; }
  return v;
}

// ------------------------------------------------------------------ ray prep
// cond[ray] = [posenc(viewdir) | appearance code (only with use_alpha_condition, models.py:206) |
// camera code]; condterm_{c,f}[ray][n] = cond . W_rgbh[256:, n] + b_rgbh[n]; with use_alpha_condition also
// alpha_ct_{c,f}[ray] = appearance code . W_alpha[256:] (modules.py:152-157).  Codes come from the embedding tables
// (glo.py:50-53) or, pre-encoded, straight from the caller (metadata_encoded, models.py:198-199, 210-211).
__global__ __launch_bounds__(128) void ray_prep_kernel(const RayPrepArgs A) {
  __shared__ float s_cond[64];
  const int ray = blockIdx.x, t = threadIdx.x;
  const float* __restrict__ params = A.params;
  const int V = A.use_viewdirs ? 3 + 6 * A.Fv : 0;
  if (t < A.R) {
    float v;
    if (t < V) {
      if (t < 3) v = A.viewdirs[3 * ray + t];
      else {
        const int q = t - 3, f = q / 6, rem = q - 6 * f, is_cos = rem / 3, c = rem - 3 * is_cos;
        float a = __fmul_rn(A.viewdirs[3 * ray + c], (float)(1 << f));
        if (is_cos) a = __fadd_rn(a, 1.57079632679489661923f);
        v = sinf(a);
      }
    } else if (t < V + A.app_feat) {
      v = A.app_codes ? A.app_codes[(size_t)ray * A.app_feat + (t - V)]
                      : params[A.app_off + (int64_t)A.app_ids[ray] * A.app_feat + (t - V)];
    } else {
      const int c = t - V - A.app_feat;
      v = A.cam_codes ? A.cam_codes[(size_t)ray * A.cam_feat + c] : params[A.cam_off + (int64_t)A.cam_ids[ray] * A.cam_feat + c];
    }
    s_cond[t] = v;
    A.cond[(size_t)ray * A.R + t] = v;
  }
  __syncthreads();
  float* ct_f = A.condterm[1];
  float a = params[A.rgbh_b[0] + t], b = ct_f ? params[A.rgbh_b[1] + t] : 0.f;
  for (int c = 0; c < A.R; ++c) {
    const float x = s_cond[c];
    a = fmaf(x, params[A.rgbh_k[0] + (int64_t)(TRUNK_W + c) * RGB_W + t], a);
    if (ct_f) b = fmaf(x, params[A.rgbh_k[1] + (int64_t)(TRUNK_W + c) * RGB_W + t], b);
  }
  A.condterm[0][(size_t)ray * RGB_W + t] = a;
  if (ct_f) ct_f[(size_t)ray * RGB_W + t] = b;
  if (t < 2 && A.alpha_ct[t]) {   // thread 0: coarse, thread 1: fine
    float s = 0.f;
    for (int c = 0; c < A.app_feat; ++c) s = fmaf(s_cond[V + c], params[A.alpha_k[t] + TRUNK_W + c], s);
    A.alpha_ct[t][ray] = s;
  }
}

void launch_ray_prep(const RayPrepArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL(ray_prep_kernel, dim3(a.B), dim3(128), 0, stream, a);
}

// ------------------------------------------------------------------ coarse sampling
__device__ __forceinline__ float coarse_z(int s, int N, float near_p, float far_p, int lindisp) {
  const float t = (s == N - 1) ? 1.0f : (float)s / (float)(N - 1);   // linspace(0,1,N)
  if (!lindisp) return near_p * (1.f - t) + far_p * t;               // model_utils.py:58
  return 1.f / (1.f / near_p * (1.f - t) + 1.f / far_p * t);         // model_utils.py:60
}

__global__ void sample_coarse_kernel(const float* __restrict__ t_rand, int B, int N, float near_p, float far_p,
                                     int stratified, int lindisp, uint64_t seed, uint64_t offset,
                                     const nrf_dynamic_scalars* __restrict__ dyn, float* __restrict__ z) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * N) return;
  if (dyn) { seed = dyn->rng_seed; offset = dyn->rng_offset; }   // graph-replayable step: the key lives on the device
  const int s = idx % N;
  const float zc = coarse_z(s, N, near_p, far_p, lindisp);
  if (!stratified) { z[idx] = zc; return; }
  // model_utils.py:62-66
  const float zl = s > 0 ? coarse_z(s - 1, N, near_p, far_p, lindisp) : zc;
  const float zu = s < N - 1 ? coarse_z(s + 1, N, near_p, far_p, lindisp) : zc;
  const float lower = s > 0 ? .5f * (zc + zl) : zc;
  const float upper = s < N - 1 ? .5f * (zu + zc) : zc;
  const float t = t_rand ? t_rand[idx] : philox_uniform(seed, offset, 0u, (uint32_t)idx);
  z[idx] = lower + (upper - lower) * t;
}

// points = origins + z_vals * directions (model_utils.py:72-73, :213-215) as an OUTPUT: with the warp field the warp kernel
// writes them; without it the MLP kernel forms them in its prologue and never stores them, so return_points
// (models.py:247-248 returns `points` whether or not the model warps) gets its own small kernel.
__global__ void sample_points_kernel(const float* __restrict__ origins, const float* __restrict__ dirs, const float* __restrict__ z,
                                     int rows, int S, float* __restrict__ out) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const int ray = r / S;
  const float zi = z[r];
#pragma unroll
  for (int c = 0; c < 3; ++c) out[3 * (size_t)r + c] = __fadd_rn(origins[3 * ray + c], __fmul_rn(zi, dirs[3 * ray + c]));
}

void launch_sample_points(const float* origins, const float* dirs, const float* z, int B, int S, float* out, hipStream_t stream) {
  const int rows = B * S;
  hipLaunchKernelGGL(sample_points_kernel, dim3((rows + 255) / 256), dim3(256), 0, stream, origins, dirs, z, rows, S, out);
}

void launch_sample_coarse(const float* t_rand, int B, int N, float near_p, float far_p, int stratified, int lindisp,
                          uint64_t seed, uint64_t offset, const nrf_dynamic_scalars* dyn, float* z, hipStream_t stream) {
  const int total = B * N;
  hipLaunchKernelGGL(sample_coarse_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, t_rand, B, N, near_p,
                     far_p, stratified, lindisp, seed, offset, dyn, z);
}

// ------------------------------------------------------------------ compositing
constexpr int MAX_E = 8;   // up to 512 samples per ray

__global__ __launch_bounds__(256) void composite_fwd_kernel(
    const float4* __restrict__ out4, const float* __restrict__ z, const float* __restrict__ dirs, int B, int S,
    int white_bkgd, int sample_at_inf, float* __restrict__ o_rgb, float* __restrict__ o_depth,
    float* __restrict__ o_med, float* __restrict__ o_acc, float* __restrict__ o_w) {
  const int lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= B) return;
  const float dx = dirs[3 * ray], dy = dirs[3 * ray + 1], dz = dirs[3 * ray + 2];
  const float dnorm = sqrtf(dx * dx + dy * dy + dz * dz);
  const float last = sample_at_inf ? 1e10f : 1e-19f;
  const float* zr = z + (size_t)ray * S;
  const float4* cr = out4 + (size_t)ray * S;
  float Tc = 1.f, Wc = 0.f;
  float r0 = 0.f, r1 = 0.f, r2 = 0.f, dep = 0.f, acc_all = 0.f, acc_nl = 0.f, med = 0.f;
  bool found = false;
  const int E = (S + 63) >> 6;
  for (int e = 0; e < E; ++e) {
    const int s = e * 64 + lane;
    const bool valid = s < S;
    float zi = 0.f, alpha = 0.f, tt = 1.f;
    float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) {
      zi = zr[s]; c = cr[s];
      const float dist = (s + 1 < S ? zr[s + 1] - zi : last) * dnorm;   // model_utils.py:104-109
      alpha = 1.0f - expf(-c.w * dist);                                // :110
      tt = 1.0f - alpha + 1e-10f;                                      // :114
    }
    const float incl = wave_incl_prod(tt, lane);
    float T = __shfl_up(incl, 1);
    if (lane == 0) T = 1.f;
    T *= Tc;
    const float w = alpha * T;                                         // :116
    const float cum = wave_incl_sum(w, lane) + Wc;
    if (valid && o_w) o_w[(size_t)ray * S + s] = w;
    r0 += w * c.x; r1 += w * c.y; r2 += w * c.z; dep += w * zi; acc_all += w;
    if (s < S - 1) acc_nl += w;
    // median depth: first sample whose cumulative weight reaches 0.5 (model_utils.py:218-263)
    const unsigned long long m = __ballot(valid && cum >= 0.5f);
    if (!found && m) {
      const int first = __ffsll((long long)m) - 1;
      med = __shfl(zi, first);
      found = true;
    }
    Tc *= __shfl(incl, 63);
    Wc = __shfl(cum, 63);
  }
  r0 = wave_sum(r0); r1 = wave_sum(r1); r2 = wave_sum(r2); dep = wave_sum(dep);
  acc_all = wave_sum(acc_all); acc_nl = wave_sum(acc_nl);
  if (lane == 0) {
    if (white_bkgd) { r0 += 1.f - acc_all; r1 += 1.f - acc_all; r2 += 1.f - acc_all; }   // :122-123
    if (o_rgb) { o_rgb[3 * ray] = r0; o_rgb[3 * ray + 1] = r1; o_rgb[3 * ray + 2] = r2; }
    if (o_depth) o_depth[ray] = dep;
    if (o_med) o_med[ray] = med;
    if (o_acc) o_acc[ray] = sample_at_inf ? acc_nl : acc_all;                             // :125-126
  }
}

void launch_composite_fwd(const float4* out4, const float* z, const float* dirs, int B, int S, int white_bkgd,
                          int sample_at_inf, float* rgb, float* depth, float* med_depth, float* acc, float* weights,
                          hipStream_t stream) {
  hipLaunchKernelGGL(composite_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, stream, out4, z, dirs, B, S, white_bkgd,
                     sample_at_inf, rgb, depth, med_depth, acc, weights);
}

// Reverse pass of volumetric_rendering + sigmoid / sigma activation + the MSE loss.
// With g_i = c_i . dL/drgb, t_i = 1-alpha_i+1e-10 and Q_i = sum_{k>i} g_k alpha_k prod_{i<j<k} t_j
// (reverse affine recurrence Q_i = g_{i+1} alpha_{i+1} + t_{i+1} Q_{i+1}; division free):
//   dL/dalpha_i = T_i (g_i - Q_i);  dL/dsigma_i = dist_i exp(-sigma_i dist_i) dL/dalpha_i;  dL/dc_i = w_i dL/drgb.
__global__ __launch_bounds__(256) void composite_bwd_kernel(const CompositeBwdArgs2 P) {
  const CompositeBwdArgs& A = P.a[blockIdx.y];   // blockIdx.y = level: both levels in one launch (kernarg-indexed, scalar loads)
  const float4* __restrict__ out4 = A.out4;
  const float* __restrict__ z = A.z;
  const float* __restrict__ dirs = A.dirs;
  const int B = A.B, S = A.S, white_bkgd = A.white_bkgd, sample_at_inf = A.sample_at_inf, sigma_act = A.sigma_act;
  const float* __restrict__ rgb_out = A.rgb_out;
  const float* __restrict__ target = A.target;
  const float* __restrict__ d_rgb = A.d_rgb;
  const float loss_scale = A.loss_scale;
  float4* __restrict__ d_raw4 = A.d_raw4;
  const int rows_pad = A.rows_pad;
  float* __restrict__ mse_ray = A.mse_ray;
  float* __restrict__ dsig_ray = A.dsig_ray;
  const int lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (blockIdx.x == 0) {   // zero the tile padding rows so they contribute nothing to any gradient
    for (int r = B * S + threadIdx.x; r < rows_pad; r += blockDim.x) d_raw4[r] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (ray >= B) return;
  float g0, g1, g2;
  if (d_rgb) {
    g0 = d_rgb[3 * ray]; g1 = d_rgb[3 * ray + 1]; g2 = d_rgb[3 * ray + 2];
    if (lane == 0 && mse_ray) mse_ray[ray] = 0.f;
  }
  else {
    const float e0 = rgb_out[3 * ray] - target[3 * ray], e1 = rgb_out[3 * ray + 1] - target[3 * ray + 1],
                e2 = rgb_out[3 * ray + 2] - target[3 * ray + 2];
    g0 = loss_scale * e0; g1 = loss_scale * e1; g2 = loss_scale * e2;   // d mean((rgb-t)^2) (training.py:172)
    if (lane == 0 && mse_ray) mse_ray[ray] = e0 * e0 + e1 * e1 + e2 * e2;   // summed in a fixed order by finish_stats_kernel
  }
  const float gsum = g0 + g1 + g2;
  const float dx = dirs[3 * ray], dy = dirs[3 * ray + 1], dz = dirs[3 * ray + 2];
  const float dnorm = sqrtf(dx * dx + dy * dy + dz * dz);
  const float last = sample_at_inf ? 1e10f : 1e-19f;
  const float* zr = z + (size_t)ray * S;
  const float4* cr = out4 + (size_t)ray * S;
  const int E = (S + 63) >> 6;
  float Tv[MAX_E], av[MAX_E], tv[MAX_E], dv[MAX_E], gv[MAX_E];
  float4 cv[MAX_E];
  float Tc = 1.f;
#pragma unroll
  for (int e = 0; e < MAX_E; ++e) {
    if (e < E) {
      const int s = e * 64 + lane;
      const bool valid = s < S;
      float alpha = 0.f, tt = 1.f, dist = 0.f;
      float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
      if (valid) {
        const float zi = zr[s]; c = cr[s];
        dist = (s + 1 < S ? zr[s + 1] - zi : last) * dnorm;
        alpha = 1.0f - expf(-c.w * dist);
        tt = 1.0f - alpha + 1e-10f;
      }
      const float incl = wave_incl_prod(tt, lane);
      float T = __shfl_up(incl, 1);
      if (lane == 0) T = 1.f;
      T *= Tc;
      Tc *= __shfl(incl, 63);
      float g = c.x * g0 + c.y * g1 + c.z * g2;
      if (white_bkgd) g -= gsum;
      Tv[e] = T; av[e] = alpha; tv[e] = tt; dv[e] = dist; gv[e] = valid ? g : 0.f; cv[e] = c;
    }
  }
  float Qin = 0.f;   // Q of the last sample of the chunk being processed
  float dsig_acc = 0.f;
#pragma unroll
  for (int e = MAX_E - 1; e >= 0; --e) {
    if (e < E) {
      const int s = e * 64 + lane;
      const bool valid = s < S;
      // suffix composition C_l = m_l o m_{l+1} o ... o m_63, m_s(Q) = g_s alpha_s + t_s Q
      float Aa = valid ? gv[e] * av[e] : 0.f, Bb = valid ? tv[e] : 1.f;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const float Ar = __shfl_down(Aa, o), Br = __shfl_down(Bb, o);
        if (lane + o < 64) { Aa = Aa + Bb * Ar; Bb = Bb * Br; }
      }
      float An = __shfl_down(Aa, 1), Bn = __shfl_down(Bb, 1);
      if (lane == 63) { An = 0.f; Bn = 1.f; }
      const float Q = An + Bn * Qin;
      Qin = __shfl(Aa, 0) + __shfl(Bb, 0) * Qin;
      if (valid) {
        const float4 c = cv[e];
        const float w = av[e] * Tv[e];
        const float dalpha = Tv[e] * (gv[e] - Q);
        const float dsigma = dv[e] * expf(-c.w * dv[e]) * dalpha;
        float4 o;
        o.x = w * g0 * c.x * (1.f - c.x);           // through sigmoid (models.py:276)
        o.y = w * g1 * c.y * (1.f - c.y);
        o.z = w * g2 * c.z * (1.f - c.z);
        o.w = sigma_act == 1 ? dsigma * (1.f - expf(-c.w)) : (c.w > 0.f ? dsigma : 0.f);   // softplus' / relu'
        d_raw4[(size_t)ray * S + s] = o;
        dsig_acc += o.w;
      }
    }
  }
  if (dsig_ray) {   // use_alpha_condition: the alpha head's per-ray input sees the sum over the ray's samples
    dsig_acc = wave_sum(dsig_acc);
    if (lane == 0) dsig_ray[ray] = dsig_acc;
  }
}

void launch_composite_bwd(const CompositeBwdArgs& a0, const CompositeBwdArgs* a1, hipStream_t stream) {
  CompositeBwdArgs2 p;
  p.a[0] = a0; p.a[1] = a1 ? *a1 : a0;
  hipLaunchKernelGGL(composite_bwd_kernel, dim3((a0.B + 3) / 4, a1 ? 2 : 1), dim3(256), 0, stream, p);
}

// ------------------------------------------------------------------ hierarchical sampling
constexpr int SF_MAXC = 256;   // max coarse samples
constexpr int SF_MAXT = 512;   // max coarse + fine samples

__global__ __launch_bounds__(256) void sample_fine_kernel(
    const float* __restrict__ z_c, const float* __restrict__ w_c, int B, int Nc, int Nf, int stratified,
    const float* __restrict__ u_in, uint64_t seed, uint64_t offset, const nrf_dynamic_scalars* __restrict__ dyn,
    float* __restrict__ z_out) {
  __shared__ float s_bins[4][SF_MAXC];
  if (dyn) { seed = dyn->rng_seed; offset = dyn->rng_offset; }
  __shared__ float s_cdf[4][SF_MAXC];
  __shared__ float s_all[4][SF_MAXT];
  __shared__ float s_fine[4][SF_MAXT];   // the fine draws, padded to a power of two with +inf for the bitonic network
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ray_raw = blockIdx.x * 4 + wv;
  const bool live = ray_raw < B;
  const int ray = live ? ray_raw : B - 1;
  const float* zr = z_c + (size_t)ray * Nc;
  const float* wr = w_c + (size_t)ray * Nc;
  float* bins = s_bins[wv]; float* cdf = s_cdf[wv]; float* all = s_all[wv];
  const int n = Nc - 1;         // bins (midpoints);  n-1 = Nc-2 interior weights (models.py:353-355)
  // pdf = (w + 1e-5) / sum ; cdf = [0, cumsum(pdf)]   (model_utils.py:153-159)
  float tot = 0.f;
  for (int m = lane; m < n - 1; m += 64) tot += wr[m + 1] + 1e-5f;
  tot = wave_sum(tot);
  float carry = 0.f;
  for (int m0 = 0; m0 < n - 1; m0 += 64) {
    const int m = m0 + lane;
    const float pdf = m < n - 1 ? (wr[m + 1] + 1e-5f) / tot : 0.f;
    const float c = wave_incl_sum(pdf, lane) + carry;
    if (m < n - 1) cdf[m + 1] = c;
    carry = __shfl(c, 63);
  }
  if (lane == 0) cdf[0] = 0.f;
  for (int m = lane; m < n; m += 64) bins[m] = .5f * (zr[m + 1] + zr[m]);
  for (int s = lane; s < Nc; s += 64) all[s] = zr[s];
  __syncthreads();
  // inverse CDF (model_utils.py:162-184), as searchsorted(cdf, u, 'right') + clamps
  for (int jx = lane; jx < Nf; jx += 64) {
    float u;
    if (stratified) u = u_in ? u_in[(size_t)ray * Nf + jx] : philox_uniform(seed, offset, 1u, (uint32_t)(ray * Nf + jx));
    else u = Nf == 1 ? 0.0f : (jx == Nf - 1) ? 1.0f : (float)jx / (float)(Nf - 1);   // linspace(0, 1, Nf); Nf = 1 -> [0]
    int lo_i = 0, hi_i = n;   // idx = #{i : cdf_i <= u}
    while (lo_i < hi_i) { const int mid = (lo_i + hi_i) >> 1; if (cdf[mid] <= u) lo_i = mid + 1; else hi_i = mid; }
    const int idx = lo_i;
    const int lo = min(max(idx - 1, 0), n - 2), hi = min(max(idx, 1), n - 1);
    const float c0 = cdf[lo], c1 = cdf[hi];
    float denom = c1 - c0;
    if (denom < 1e-5f) denom = 1.f;
    const float t = (u - c0) / denom;
    all[Nc + jx] = bins[lo] + t * (bins[hi] - bins[lo]);
  }
  __syncthreads();
  // sort(concat(z_coarse, z_samples)) (model_utils.py:213).  The fine draws are unordered (u is i.i.d. uniform, not
  // stratified: model_utils.py:196-198) but z_coarse is already ascending, so: bitonic-sort the Nf fine values in LDS
  // (log^2 passes; rounds 1-2 counted ranks over all (Nc+Nf)^2 pairs -- 118 us at 256+256), then place both lists by
  // binary-search ranks (equal values: coarse first; equal values are indistinguishable in the output anyway).
  // z_coarse can fail to be ascending only through a last-ulp rounding of lower + (upper - lower) * r against the next
  // stratum's lower edge; the block checks and falls back to the rank count in that case.
  const int Ntot = Nc + Nf;
  bool unsorted = false;
  for (int a = lane; a + 1 < Nc; a += 64) unsorted = unsorted || all[a] > all[a + 1];
  if (!__syncthreads_or(unsorted ? 1 : 0)) {
    float* fine = s_fine[wv];
    int P = 64;
    while (P < Nf) P <<= 1;
    for (int i = lane; i < P; i += 64) fine[i] = i < Nf ? all[Nc + i] : __builtin_inff();
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = lane; i < P; i += 64) {
          const int l = i ^ j;
          if (l > i) {
            const float a = fine[i], b = fine[l];
            if ((a > b) == ((i & k) == 0)) { fine[i] = b; fine[l] = a; }
          }
        }
        __syncthreads();
      }
    for (int i = lane; i < Nf; i += 64) {     // fine value i: after the coarse values <= it
      const float v = fine[i];
      int lo_i = 0, hi_i = Nc;
      while (lo_i < hi_i) { const int mid = (lo_i + hi_i) >> 1; if (all[mid] <= v) lo_i = mid + 1; else hi_i = mid; }
      if (live) z_out[(size_t)ray * Ntot + i + lo_i] = v;
    }
    for (int a = lane; a < Nc; a += 64) {     // coarse value a: after the fine values < it
      const float v = all[a];
      int lo_i = 0, hi_i = Nf;
      while (lo_i < hi_i) { const int mid = (lo_i + hi_i) >> 1; if (fine[mid] < v) lo_i = mid + 1; else hi_i = mid; }
      if (live) z_out[(size_t)ray * Ntot + a + lo_i] = v;
    }
    return;
  }
  for (int a = lane; a < Ntot; a += 64) {     // stable rank counting
    const float v = all[a];
    int rank = 0;
    for (int k = 0; k < Ntot; ++k) { const float o = all[k]; rank += (o < v || (o == v && k < a)) ? 1 : 0; }
    if (live) z_out[(size_t)ray * Ntot + rank] = v;
  }
}

void launch_sample_fine(const float* z_c, const float* w_c, int B, int Nc, int Nf, int stratified, const float* u,
                        uint64_t seed, uint64_t offset, const nrf_dynamic_scalars* dyn, float* z_out, hipStream_t stream) {
  hipLaunchKernelGGL(sample_fine_kernel, dim3((B + 3) / 4), dim3(256), 0, stream, z_c, w_c, B, Nc, Nf, stratified, u,
                     seed, offset, dyn, z_out);
}

// ------------------------------------------------------------------ small gradient pieces
// dW_rgbh[256+c][n] = sum_ray cond[ray][c] * dray[ray][n].  One block per condition column c:
// 8 ray groups x 128 outputs, 4 independent accumulators per thread, LDS tree over the groups.
__global__ __launch_bounds__(1024) void cond_wgrad_kernel(const float* __restrict__ cond, const float* __restrict__ dray0,
                                                          const float* __restrict__ dray1, int B, int R,
                                                          float* __restrict__ dst0, float* __restrict__ dst1) {
  __shared__ float red[8][RGB_W];
  const float* __restrict__ dray = blockIdx.y ? dray1 : dray0;   // blockIdx.y = level: both levels in one launch
  float* __restrict__ dst = blockIdx.y ? dst1 : dst0;
  const int c = blockIdx.x, n = threadIdx.x & 127, g = threadIdx.x >> 7;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int ray = g;
  for (; ray + 24 < B; ray += 32) {
    a0 = fmaf(cond[(size_t)ray * R + c], dray[(size_t)ray * RGB_W + n], a0);
    a1 = fmaf(cond[(size_t)(ray + 8) * R + c], dray[(size_t)(ray + 8) * RGB_W + n], a1);
    a2 = fmaf(cond[(size_t)(ray + 16) * R + c], dray[(size_t)(ray + 16) * RGB_W + n], a2);
    a3 = fmaf(cond[(size_t)(ray + 24) * R + c], dray[(size_t)(ray + 24) * RGB_W + n], a3);
  }
  for (; ray < B; ray += 8) a0 = fmaf(cond[(size_t)ray * R + c], dray[(size_t)ray * RGB_W + n], a0);
  red[g][n] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (g == 0) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += red[q][n];
    dst[(size_t)c * RGB_W + n] = s;
  }
}

void launch_cond_wgrad(const float* cond, const float* dray0, const float* dray1, int B, int R, float* dst0, float* dst1,
                       hipStream_t stream) {
  if (R > 0)
    hipLaunchKernelGGL(cond_wgrad_kernel, dim3(R, dray1 ? 2 : 1), dim3(1024), 0, stream, cond, dray0, dray1, B, R, dst0, dst1);
}

// dL/d(GLO code) of the rgb-branch conditions -> scatter-add into the embedding-table gradients
// (transpose of the nn.Embed gathers of models.py:197-214).  One WAVE per ray (lane holds dray[n], n = lane and lane + 64):
// d cond[c] = sum_n dray[ray][n] * W_rgbh[256 + V + c][n], one shuffle reduction per code entry; the 16 rays of a block then
// merge rows that share an id (a rig has TWO camera ids: one atomic per ray and entry would be ~B/2 same-address atomics per
// table row) -- the first ray of the block with an id owns that id's sum.  blockIdx.y = level.
constexpr int CEG_RAYS = 16, CEG_MAXC = 64;   // nrf_create bounds the rgb condition width (viewdirs + codes) by 64
struct CondEmbedArgs {
  const float* params; const float* dray[2]; const int32_t* app_ids; const int32_t* cam_ids;
  int B, V, app_feat, cam_feat; int64_t app_off, cam_off, rgbh_k[2]; float* grad;
};
__global__ __launch_bounds__(64 * CEG_RAYS) void cond_embed_grad_kernel(const CondEmbedArgs A) {
  __shared__ float tot_s[CEG_RAYS][CEG_MAXC];
  __shared__ int ida_s[CEG_RAYS], idc_s[CEG_RAYS];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int ray = blockIdx.x * CEG_RAYS + w;
  const int lv = blockIdx.y;
  const int C = A.app_feat + A.cam_feat;
  if (ray < A.B) {
    const float* __restrict__ dray = A.dray[lv];
    const float d0 = dray[(size_t)ray * RGB_W + lane], d1 = dray[(size_t)ray * RGB_W + 64 + lane];
    const float* __restrict__ wt = A.params + A.rgbh_k[lv] + (int64_t)(TRUNK_W + A.V) * RGB_W;
    for (int c = 0; c < C; ++c) {
      const float tot = wave_sum(fmaf(d0, wt[(int64_t)c * RGB_W + lane], d1 * wt[(int64_t)c * RGB_W + 64 + lane]));
      if (lane == 0) tot_s[w][c] = tot;
    }
  }
  if (lane == 0) {
    ida_s[w] = ray < A.B && A.app_feat ? A.app_ids[ray] : -1;
    idc_s[w] = ray < A.B && A.cam_feat ? A.cam_ids[ray] : -1;
  }
  __syncthreads();
  // thread (r = ray of the block, c = code entry): leader election per table
  for (int t = threadIdx.x; t < CEG_RAYS * C; t += blockDim.x) {
    const int r = t / C, c = t - r * C;
    const bool app = c < A.app_feat;
    const int* ids = app ? ida_s : idc_s;
    const int id = ids[r];
    if (id < 0) continue;
    bool leader = true;
    for (int e = 0; e < r; ++e) leader = leader && ids[e] != id;
    if (!leader) continue;
    float sm = 0.f;
    for (int e = r; e < CEG_RAYS; ++e) sm += ids[e] == id ? tot_s[e][c] : 0.f;
    if (app) atomicAdd(A.grad + A.app_off + (int64_t)id * A.app_feat + c, sm);
    else atomicAdd(A.grad + A.cam_off + (int64_t)id * A.cam_feat + (c - A.app_feat), sm);
  }
}

void launch_cond_embed_grad(const float* params, const float* dray0, const float* dray1, const int32_t* app_ids,
                            const int32_t* cam_ids, int B, int V, int app_feat, int64_t app_off, int cam_feat, int64_t cam_off,
                            int64_t rgbh_k0, int64_t rgbh_k1, float* grad, hipStream_t stream) {
  if (app_feat + cam_feat <= 0) return;
  CondEmbedArgs a;
  a.params = params; a.dray[0] = dray0; a.dray[1] = dray1 ? dray1 : dray0; a.app_ids = app_ids; a.cam_ids = cam_ids;
  a.B = B; a.V = V; a.app_feat = app_feat; a.cam_feat = cam_feat; a.app_off = app_off; a.cam_off = cam_off;
  a.rgbh_k[0] = rgbh_k0; a.rgbh_k[1] = rgbh_k1; a.grad = grad;
  hipLaunchKernelGGL(cond_embed_grad_kernel, dim3((B + CEG_RAYS - 1) / CEG_RAYS, dray1 ? 2 : 1), dim3(64 * CEG_RAYS), 0, stream, a);
}

// use_alpha_condition: the alpha head is Dense([bottleneck, appearance code] -> 1) (modules.py:152-157).  Its
// bottleneck rows are a wgrad group; here the code rows  dW[256 + a] = sum_ray code[ray][a] * dsig[ray]  and the
// gradient of the codes through the head,  d code[ray][a] = W[256 + a] * dsig[ray]  (scatter-add into the table).
// One block per code channel a.
__global__ __launch_bounds__(256) void alpha_cond_grad_kernel(const float* __restrict__ params, const float* __restrict__ cond,
                                                              const float* __restrict__ dsig, const int32_t* __restrict__ app_ids,
                                                              int B, int R, int V, int app_feat, int64_t app_off, int64_t alpha_k,
                                                              float* __restrict__ grad) {
  __shared__ float red[4];
  const int a = blockIdx.x, t = threadIdx.x;
  const float w = params[alpha_k + TRUNK_W + a];
  float s = 0.f;
  for (int ray = t; ray < B; ray += 256) {
    const float d = dsig[ray];
    s = fmaf(cond[(size_t)ray * R + V + a], d, s);
    if (app_ids && d != 0.f) atomicAdd(grad + app_off + (int64_t)app_ids[ray] * app_feat + a, w * d);
  }
  s = wave_sum(s);
  if ((t & 63) == 0) red[t >> 6] = s;
  __syncthreads();
  if (t == 0) grad[alpha_k + TRUNK_W + a] = (red[0] + red[1]) + (red[2] + red[3]);
}

void launch_alpha_cond_grad(const float* params, const float* cond, const float* dsig_ray, const int32_t* app_ids, int B, int R,
                            int V, int app_feat, int64_t app_off, int64_t alpha_k, float* grad, hipStream_t stream) {
  if (app_feat > 0)
    hipLaunchKernelGGL(alpha_cond_grad_kernel, dim3(app_feat), dim3(256), 0, stream, params, cond, dsig_ray, app_ids, B, R, V,
                       app_feat, app_off, alpha_k, grad);
}

__global__ __launch_bounds__(64) void finish_stats_kernel(const StatsArgs A) {
  // per-ray squared errors of the two levels (composite_bwd_kernel) -> their sums, in a fixed order (no atomics: two runs of
  // the same step report the same loss bits)
  float sc = 0.f, sf = 0.f;
  for (int r = threadIdx.x; r < A.B; r += 64) {
    sc += A.mse_ray[r];
    if (A.nlevels > 1) sf += A.mse_ray[A.B + r];
  }
  sc = wave_sum(sc); sf = wave_sum(sf);
  float el[5] = {0.f, 0.f, 0.f, 0.f, 0.f};   // elastic_kernel's per-workgroup partials, summed in a fixed order as well
  if (A.el_part) {
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      for (int w = threadIdx.x; w < A.el_nwg; w += 64) el[q] += A.el_part[(size_t)q * A.el_nwg + w];
      el[q] = wave_sum(el[q]);
    }
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float* stats = A.stats;
    const float B = (float)A.B;
    const float mc = sc / (3.f * B), mf = sf / (3.f * B);
    const float bgl = A.bg_sum ? A.bg_sum[0] / (float)A.bgN : 0.f;
    const float ell = A.el_part ? el[0] / B : 0.f;          // sum over samples, mean over rays (training.py:194)
    const float wrc = A.wr_sums ? A.wr_sums[0] / B : 0.f, wrf = A.wr_sums ? A.wr_sums[2] / B : 0.f;
    stats[0] = mc; stats[1] = mf;
    stats[2] = -10.f * logf(mc) / logf(10.f);   // utils.compute_psnr (utils.py:94-103)
    stats[3] = -10.f * logf(mf) / logf(10.f);
    const float el_weight = A.dyn ? A.dyn->elastic_loss_weight : A.el_weight;
    stats[4] = mc + mf + A.bg_weight * bgl + el_weight * ell + A.wr_weight * (wrc + wrf);   // training.py:261 (+ :197, :212, :257-258)
    stats[5] = bgl;                               // stats['background_loss'] (training.py:259)
    stats[6] = ell;                               // stats['loss/elastic']
    stats[7] = A.el_part ? el[1] / (float)A.el_rows : 0.f;   // stats['residual/elastic'] (training.py:196)
    stats[8] = wrc; stats[9] = wrf;               // stats['loss/warp_reg'] coarse / fine (training.py:210)
    stats[10] = A.wr_sums ? A.wr_sums[1] / B : 0.f;   // stats['residual/warp_reg'] (training.py:211)
    stats[11] = A.wr_sums ? A.wr_sums[3] / B : 0.f;
    const float jr = (float)(A.el_jac_rows > 0 ? A.el_jac_rows : 1);
    stats[12] = A.el_part ? el[2] / jr : 0.f;  // metric/jacobian_det, _div, _curl (training.py:214-222): mean over all coarse samples
    stats[13] = A.el_part ? el[3] / jr : 0.f;
    stats[14] = A.el_part ? el[4] / jr : 0.f;
    stats[15] = 0.f;
  }
}

void launch_finish_stats(const StatsArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL(finish_stats_kernel, dim3(1), dim3(64), 0, stream, a);
}

// ------------------------------------------------------------------ warp regulariser
// training.py:199-212 for one level: residual = |points - warped_points|^2 at the sample of
// model_utils.compute_depth_index(stop_gradient(weights)) (first sample whose cumulative weight reaches 0.5, sample 0
// if none), loss = general_loss_with_squared_residual(residual, alpha, scale), mean over rays.  One wave per ray;
// d loss / d warped point is ADDED to the selected row of d_points (written before by the NeRF MLP's dgrad).
__global__ __launch_bounds__(256) void warp_reg_kernel(const float* __restrict__ w, const float* __restrict__ pts,
                                                       const float* __restrict__ warped, int B, int S, float alpha, float scale,
                                                       float gscale, float* __restrict__ d_points, float* __restrict__ sums) {
  const int lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= B) return;
  float carry = 0.f;
  int idx = 0;
  bool found = false;
  for (int s0 = 0; s0 < S; s0 += 64) {
    const int s = s0 + lane;
    const float cum = wave_incl_sum(s < S ? w[(size_t)ray * S + s] : 0.f, lane) + carry;
    const unsigned long long m = __ballot(s < S && cum >= 0.5f);
    if (!found && m) { idx = s0 + __ffsll((long long)m) - 1; found = true; }
    carry = __shfl(cum, 63);
  }
  if (lane == 0) {
    const size_t row = (size_t)ray * S + idx;
    float r[3], q = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) { r[c] = warped[3 * row + c] - pts[3 * row + c]; q += r[c] * r[c]; }
    float rho, drho;
    general_loss_sq(q, alpha, scale, rho, drho);
#pragma unroll
    for (int c = 0; c < 3; ++c) d_points[3 * row + c] += gscale * drho * 2.f * r[c];
    atomicAdd(sums, rho);
    atomicAdd(sums + 1, sqrtf(q));
  }
}

void launch_warp_reg(const float* weights, const float* points, const float* warped, int B, int S, float alpha, float scale,
                     float gscale, float* d_points, float* sums, hipStream_t stream) {
  hipLaunchKernelGGL(warp_reg_kernel, dim3((B + 3) / 4), dim3(256), 0, stream, weights, points, warped, B, S, alpha, scale, gscale,
                     d_points, sums);
}

// ------------------------------------------------------------------ elastic 'median' reduce
// coef[ray][s] = 1 at model_utils.compute_depth_index(weights) (first sample whose cumulative weight reaches
// 0.5, sample 0 if none does: argmax of an all-zero mask, model_utils.py:218-246), else 0.
__global__ __launch_bounds__(256) void median_coef_kernel(const float* __restrict__ w, int B, int S, float* __restrict__ coef) {
  const int lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= B) return;
  float carry = 0.f;
  int idx = 0;
  bool found = false;
  for (int s0 = 0; s0 < S; s0 += 64) {
    const int s = s0 + lane;
    const float cum = wave_incl_sum(s < S ? w[(size_t)ray * S + s] : 0.f, lane) + carry;
    const unsigned long long m = __ballot(s < S && cum >= 0.5f);
    if (!found && m) { idx = s0 + __ffsll((long long)m) - 1; found = true; }
    carry = __shfl(cum, 63);
  }
  for (int s = lane; s < S; s += 64) coef[(size_t)ray * S + s] = s == idx ? 1.f : 0.f;
}

void launch_median_coef(const float* weights, int B, int S, float* coef, hipStream_t stream) {
  hipLaunchKernelGGL(median_coef_kernel, dim3((B + 3) / 4), dim3(256), 0, stream, weights, B, S, coef);
}

// ------------------------------------------------------------------ background regulariser
// loss_i = general_loss_with_squared_residual(|x'_i - x_i|^2, alpha, scale) (utils.py:264-331, every branch:
// general_loss.h);  d rho/d x' = rho'(q) * 2 (x' - x).
// d_points = weight/N * d rho/d x' (0 on the tile padding rows);  loss_sum += sum_i rho_i.
__global__ __launch_bounds__(256) void background_loss_kernel(const float* __restrict__ pts, const float* __restrict__ warped,
                                                              int N, int rows_pad, float alpha, float scale, float gscale,
                                                              float* __restrict__ d_points, float* __restrict__ loss_sum) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float rho = 0.f;
  if (i < rows_pad) {
    float g[3] = {0.f, 0.f, 0.f};
    if (i < N) {
      float r[3], q = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) { r[c] = warped[3 * i + c] - pts[3 * i + c]; q += r[c] * r[c]; }
      float drho;
      general_loss_sq(q, alpha, scale, rho, drho);
#pragma unroll
      for (int c = 0; c < 3; ++c) g[c] = gscale * drho * 2.f * r[c];
    }
    d_points[3 * i] = g[0]; d_points[3 * i + 1] = g[1]; d_points[3 * i + 2] = g[2];
  }
  rho = wave_sum(rho);
  if ((threadIdx.x & 63) == 0 && rho != 0.f) atomicAdd(loss_sum, rho);
}

void launch_background_loss(const float* points, const float* warped, int N, int rows_pad, float alpha, float scale,
                            float weight, float* d_points, float* loss_sum, hipStream_t stream) {
  hipLaunchKernelGGL(background_loss_kernel, dim3((rows_pad + 255) / 256), dim3(256), 0, stream, points, warped, N, rows_pad,
                     alpha, scale, weight / (float)N, d_points, loss_sum);
}

// training.compute_background_loss's two draws (training.py:121-126), one thread per point: warp id = choices[floor(U n)]
// (random.choice over model.warp_ids; Philox stream 4), x += noise_std * N(0, 1) per coordinate (stream 5, element 3 i + c).
__global__ __launch_bounds__(256) void background_draw_kernel(const float* __restrict__ pts, int N, const int32_t* __restrict__ choices,
                                                              int nchoices, float noise_std, uint64_t seed, uint64_t offset,
                                                              const nrf_dynamic_scalars* __restrict__ dyn, float* __restrict__ out_pts,
                                                              int32_t* __restrict__ out_ids) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  if (dyn) { seed = dyn->rng_seed; offset = dyn->rng_offset; }
  const int k = min((int)(philox_uniform(seed, offset, 4u, (uint32_t)i) * (float)nchoices), nchoices - 1);
  out_ids[i] = choices[k];
#pragma unroll
  for (int c = 0; c < 3; ++c) out_pts[3 * i + c] = pts[3 * i + c] + noise_std * philox_normal(seed, offset, 5u, (uint32_t)(3 * i + c));
}

void launch_background_draw(const float* points, int N, const int32_t* choices, int nchoices, float noise_std, uint64_t seed,
                            uint64_t offset, const nrf_dynamic_scalars* dyn, float* out_points, int32_t* out_ids, hipStream_t stream) {
  hipLaunchKernelGGL(background_draw_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, points, N, choices, nchoices, noise_std, seed,
                     offset, dyn, out_points, out_ids);
}

// ------------------------------------------------------------------ Adam
__global__ void adam_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                            const float* __restrict__ g, int64_t n, float lr, float b1, float omb1, float b2,
                            float omb2, float eps, float c1, float c2, float gscale) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * gscale;
    const float mi = b1 * m[i] + omb1 * gi;
    const float vi = b2 * v[i] + omb2 * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] = p[i] - lr * (mi / c1) / (sqrtf(vi / c2) + eps);
  }
}

__global__ void dynamic_write_kernel(nrf_dynamic_scalars* __restrict__ dst, const nrf_dynamic_scalars v) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *dst = v;
}
void launch_dynamic_write(nrf_dynamic_scalars* dst, const nrf_dynamic_scalars& v, hipStream_t stream) {
  hipLaunchKernelGGL(dynamic_write_kernel, dim3(1), dim3(64), 0, stream, dst, v);
}

// the same update with lr / bias corrections / grad scale read from the device-resident step scalars (graph-replayable)
__global__ void adam_dynamic_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                    const float* __restrict__ g, int64_t n, float b1, float omb1, float b2, float omb2, float eps,
                                    const nrf_dynamic_scalars* __restrict__ dyn) {
  const float lr = dyn->learning_rate, c1 = dyn->adam_c1, c2 = dyn->adam_c2, gscale = dyn->grad_scale;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * gscale;
    const float mi = b1 * m[i] + omb1 * gi;
    const float vi = b2 * v[i] + omb2 * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] = p[i] - lr * (mi / c1) / (sqrtf(vi / c2) + eps);
  }
}

void launch_adam_dynamic(float* p, float* m, float* v, const float* g, int64_t n, double b1, double b2, double eps,
                         const nrf_dynamic_scalars* dyn, hipStream_t stream) {
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adam_dynamic_kernel, dim3(blocks), dim3(256), 0, stream, p, m, v, g, n, (float)b1, (float)(1.0 - b1), (float)b2,
                     (float)(1.0 - b2), (float)eps, dyn);
}

void launch_adam(float* p, float* m, float* v, const float* g, int64_t n, double lr, double b1, double b2, double eps,
                 int64_t step, double gscale, hipStream_t stream) {
  const double t = (double)step + 1.0;
  const float c1 = (float)(1.0 - pow(b1, t)), c2 = (float)(1.0 - pow(b2, t));
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, stream, p, m, v, g, n, (float)lr, (float)b1,
                     (float)(1.0 - b1), (float)b2, (float)(1.0 - b2), (float)eps, c1, c2, (float)gscale);
}

// One launch instead of a hipMemsetAsync per buffer (round 2: seven fillBufferAligned launches per training step): range r =
// blockIdx.y; pointers are 16-byte aligned, the tail past the last float4 is written by scalar stores.
__global__ __launch_bounds__(256) void zero_ranges_kernel(const ZeroArgs A) {
  float* __restrict__ p = A.p[blockIdx.y];
  const long long n = A.n[blockIdx.y];
  const long long n4 = n >> 2;
  float4* __restrict__ p4 = reinterpret_cast<float4*>(p);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x)
    p4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) p[4 * n4 + threadIdx.x] = 0.f;
}

void launch_zero_ranges(const ZeroArgs& a, hipStream_t stream) {
  if (a.count <= 0) return;
  hipLaunchKernelGGL(zero_ranges_kernel, dim3(64, a.count), dim3(256), 0, stream, a);
}

namespace {
__global__ __launch_bounds__(256) void embed_kernel(const EmbedDesc* __restrict__ descs, const float* __restrict__ src,
                                                    float* __restrict__ dst, int to_internal) {
  const EmbedDesc d = descs[blockIdx.y];
  if (d.ext_off < 0) {   // internal-only identity layer (the bottleneck of a model without conditions): ones on the diagonal
    if (to_internal)
      for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < d.rows; i += (long)gridDim.x * blockDim.x)
        dst[d.int_off + i * d.int_cols + i] = 1.f;
    return;
  }
  const long n = (long)d.rows * d.ext_cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / d.ext_cols), c = (int)(i - (long)r * d.ext_cols);
    const long ii = d.int_off + (long)(r < d.split ? r : r + d.shift) * d.int_cols + c;
    if (to_internal) dst[ii] = src[d.ext_off + i];
    else dst[d.ext_off + i] = src[ii];
  }
}
}  // namespace

void launch_embed(const EmbedDesc* descs, int ndesc, const float* src, float* dst, bool to_internal, hipStream_t stream) {
  if (ndesc <= 0) return;
  embed_kernel<<<dim3(64, ndesc), 256, 0, stream>>>(descs, src, dst, to_internal ? 1 : 0);
}

}  // namespace nrf
