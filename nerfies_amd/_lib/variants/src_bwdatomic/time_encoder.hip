// modules.TimeEncoder as the warp field's metadata encoder (warp_metadata_encoder_type = 'time'):
//   code = MLP(depth 6, width 64, skip at 4, output G)(AnnealedSinusoidalEncoder(num_freqs)(time, time_alpha))
// Replaces (reference, /root/reference/nerfies): modules.py:297-322 (TimeEncoder), modules.py:231-294 (the annealed
// posenc of the scalar time stamp), warping.py:256-259, 311-313 (its use by SE3Field / TranslationField),
// models.py:252-254 (metadata['time'] instead of metadata['warp']).
//
// The encoder is evaluated ONCE PER RAY (every sample of a ray shares its time stamp; the reference broadcasts the
// stamp to the samples and evaluates B*S times): one 64-lane wave per ray, lane = hidden unit, the layer input is
// broadcast lane by lane with v_readlane -- 6 x 64 x 64 MACs per ray, bandwidth- and latency-trivial next to the warp
// trunk.  Its output (B, G) takes the place of the gathered GLO rows: the warp kernels read it as a table indexed by the
// ray number, and scatter the code gradient back the same way.
#include "nrf_internal.h"

namespace nrf {

namespace {

__device__ __forceinline__ float lane_bcast(float v, int k) { return __shfl(v, k); }

// posenc of the time stamp: [t, w_0 sin(t), w_0 sin(t + pi/2), w_1 sin(2t), ...]  (modules.py:213-228, 256-270)
__device__ __forceinline__ float time_feature(float t, int k, int F, float alpha) {
  if (k == 0) return t;
  const int f = (k - 1) >> 1, is_cos = (k - 1) & 1;
  if (f >= F) return 0.f;
  const float pi = 3.14159265358979323846f;
  const float cl = fminf(fmaxf(alpha - (float)f, 0.f), 1.f);
  const float wdw = 0.5f * (1.f + cosf(__fadd_rn(__fmul_rn(pi, cl), pi)));   // cosine_easing_window (modules.py:274-294)
  float a = __fmul_rn(t, (float)(1 << f));
  if (is_cos) a = __fadd_rn(a, 1.57079632679489661923f);
  return wdw * sinf(a);
}

}  // namespace

// grid: ceil(B / 4) blocks of 4 waves.  st_h: [B][6][64] post-ReLU activations (training), st_in: [B][TIME_MAX_IN].
__global__ __launch_bounds__(256) void time_encoder_fwd_kernel(const TimeEncArgs A) {
  const int lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= A.B) return;
  const float* __restrict__ prm = A.params;
  const float tin = lane < A.Tin ? time_feature(A.time[ray], lane, A.F, A.dyn ? A.dyn->time_alpha : A.alpha) : 0.f;   // lane k holds input feature k
  if (A.st_in && lane < TIME_MAX_IN) A.st_in[(size_t)ray * TIME_MAX_IN + lane] = tin;
  float h = 0.f;
  for (int l = 0; l < TIME_DEPTH; ++l) {
    const float* __restrict__ W = prm + A.po.k[l];
    float pre = prm[A.po.b[l] + lane];
    if (l == 0) {
      for (int k = 0; k < A.Tin; ++k) pre = fmaf(lane_bcast(tin, k), W[k * TIME_W + lane], pre);
    } else {
#pragma unroll 8
      for (int k = 0; k < TIME_W; ++k) pre = fmaf(lane_bcast(h, k), W[k * TIME_W + lane], pre);
      if (l == TIME_SKIP)   // skip concat [h, inputs] (modules.py:47-48)
        for (int k = 0; k < A.Tin; ++k) pre = fmaf(lane_bcast(tin, k), W[(TIME_W + k) * TIME_W + lane], pre);
    }
    h = pre > 0.f ? pre : 0.f;
    if (A.st_h) A.st_h[((size_t)ray * TIME_DEPTH + l) * TIME_W + lane] = h;
  }
  // logit layer: code[g] = b[g] + sum_k h[k] W[k][g]  (no activation)
  const float* __restrict__ Wl = prm + A.po.lk;
  float c = lane < A.G ? prm[A.po.lb + lane] : 0.f;
  for (int k = 0; k < TIME_W; ++k) {
    const float hk = lane_bcast(h, k);
    if (lane < A.G) c = fmaf(hk, Wl[k * A.G + lane], c);
  }
  if (lane < A.G) A.codes[(size_t)ray * A.G + lane] = c;
}

void launch_time_encoder_fwd(const TimeEncArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL(time_encoder_fwd_kernel, dim3((a.B + 3) / 4), dim3(256), 0, stream, a);
}

// Reverse pass per ray: d code -> d pre of every layer (stored for the weight gradients).
__global__ __launch_bounds__(256) void time_encoder_bwd_kernel(const TimeEncArgs A) {
  const int lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= A.B) return;
  const float* __restrict__ prm = A.params;
  const float* __restrict__ Wl = prm + A.po.lk;
  const float dc = lane < A.G ? A.d_codes[(size_t)ray * A.G + lane] : 0.f;
  // d h6[k] = sum_g d code[g] W_logit[k][g]
  float dh = 0.f;
  for (int g = 0; g < A.G; ++g) dh = fmaf(lane_bcast(dc, g), Wl[lane * A.G + g], dh);
  for (int l = TIME_DEPTH - 1; l >= 0; --l) {
    const float hl = A.st_h[((size_t)ray * TIME_DEPTH + l) * TIME_W + lane];
    const float dpre = hl > 0.f ? dh : 0.f;
    A.st_dpre[((size_t)ray * TIME_DEPTH + l) * TIME_W + lane] = dpre;
    if (l == 0) break;
    // d h_{l}[k] (the input of layer l, = output of layer l-1) = sum_n dpre_l[n] W_l[k][n]
    const float* __restrict__ W = prm + A.po.k[l] + (size_t)lane * TIME_W;
    float acc = 0.f;
#pragma unroll 8
    for (int n = 0; n < TIME_W; ++n) acc = fmaf(lane_bcast(dpre, n), W[n], acc);
    dh = acc;
  }
}

void launch_time_encoder_bwd(const TimeEncArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL(time_encoder_bwd_kernel, dim3((a.B + 3) / 4), dim3(256), 0, stream, a);
}

// Weight / bias gradients: one block per (layer, input row k) -- row k of dW_l = sum_ray x_l[ray][k] dpre_l[ray][:].
// blockIdx.x enumerates: layer l rows [0, in_l) then one extra "row" per layer for the bias; the logit layer last.
__global__ __launch_bounds__(256) void time_encoder_wgrad_kernel(const TimeEncArgs A, float* __restrict__ grad) {
  __shared__ float red[4][TIME_W];
  const int n = threadIdx.x & 63, grp = threadIdx.x >> 6;
  int idx = blockIdx.x, l = 0, in_l = 0;
  for (; l <= TIME_DEPTH; ++l) {   // l == TIME_DEPTH: the logit layer
    in_l = l == 0 ? A.Tin : l == TIME_SKIP ? TIME_W + A.Tin : TIME_W;
    if (idx < in_l + 1) break;
    idx -= in_l + 1;
  }
  if (l > TIME_DEPTH) return;
  const bool logit = l == TIME_DEPTH;
  const int ncols = logit ? A.G : TIME_W;
  const bool bias_row = idx == in_l;
  float s = 0.f;
  for (int ray = grp; ray < A.B; ray += 4) {
    // x: the layer's input feature idx of this ray
    float x = 1.f;
    if (!bias_row) {
      if (l == 0) x = A.st_in[(size_t)ray * TIME_MAX_IN + idx];
      else if (l == TIME_SKIP && idx >= TIME_W) x = A.st_in[(size_t)ray * TIME_MAX_IN + (idx - TIME_W)];
      else x = A.st_h[((size_t)ray * TIME_DEPTH + (l - 1)) * TIME_W + idx];
    }
    const float d = n < ncols ? (logit ? A.d_codes[(size_t)ray * A.G + n] : A.st_dpre[((size_t)ray * TIME_DEPTH + l) * TIME_W + n]) : 0.f;
    s = fmaf(x, d, s);
  }
  red[grp][n] = s;
  __syncthreads();
  if (grp == 0 && n < ncols) {
    const float t = (red[0][n] + red[1][n]) + (red[2][n] + red[3][n]);
    const int64_t base = bias_row ? (logit ? A.po.lb : A.po.b[l]) : (logit ? A.po.lk : A.po.k[l]) + (int64_t)idx * ncols;
    grad[base + n] = t;
  }
}

void launch_time_encoder_wgrad(const TimeEncArgs& a, float* grad, hipStream_t stream) {
  int blocks = 0;
  for (int l = 0; l <= TIME_DEPTH; ++l) blocks += (l == 0 ? a.Tin : l == TIME_SKIP ? TIME_W + a.Tin : TIME_W) + 1;
  hipLaunchKernelGGL(time_encoder_wgrad_kernel, dim3(blocks), dim3(256), 0, stream, a, grad);
}

}  // namespace nrf
