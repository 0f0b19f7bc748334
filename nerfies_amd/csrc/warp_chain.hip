// Fused SE(3) warp field for gfx950: annealed posenc + GLO code -> 6x128 trunk (skip at 4) ->
// w, v heads -> exp_se3 applied to the sample point; forward and data-gradient passes.
//
// Replaces (reference, /root/reference/nerfies):
//   modules.AnnealedSinusoidalEncoder   modules.py:231-294  (fused into the tile prologue)
//   glo.GloEncoder                      glo.py:22-53        (one gather per row from the table)
//   warping.SE3Field.warp / __call__    warping.py:322-389
//   rigid_body.exp_se3 / exp_so3 / skew rigid_body.py:21-97 (closed form)
//
// Same tiling as the NeRF MLP chain (mlp_chain.hip): one workgroup = 4 waves = one 64-row tile,
// activations feature-major in LDS ([128][64] swizzled, 32 KiB) next to the trunk input tile
// ([PKw][64], <= 16 KiB) -> three workgroups per CU; each wave owns 64 rows x 32 columns (2 MFMA
// row blocks x 1 column block), weights stream from L2 in B-fragment order.  The heads (128 -> 3+3) and exp_se3 run on
// the VALU in the epilogue, one row per thread.
#include "chain_common.h"

namespace nrf {

// small_part layout (floats): db_trunk[6][128] | db_w[3] | db_v[3]
constexpr int WSP_DB_TRUNK = 0, WSP_DB_W = 768, WSP_DB_V = 771;

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
  return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

// Coefficients of the closed form of exp_se3 applied to a point, as functions of
// t2 = |w|^2:  A = sin t / t,  B = (1 - cos t) / t^2,  C = (t - sin t) / t^3  and their
// derivatives  dA/dw = Ab w,  dB/dw = Bb w,  dC/dw = Cb w  with
//   Ab = C - B... (= (t cos t - sin t)/t^3),  Bb = (A - 2B)/t^2,  Cb = (B - 3C)/t^2.
// The reference evaluates the un-simplified normalised-axis form in fp32 (rigid_body.py:54-89),
// whose 1-cos / t-sin terms cancel catastrophically for small angles and are NaN at t = 0; here
// small angles use the Taylor series, so the result tracks the exact value to fp32 rounding.
struct Se3Coef { float A, B, C, Ab, Bb, Cb; };
__device__ __forceinline__ Se3Coef se3_coef(float t2) {
  Se3Coef c;
  if (t2 < 0.04f) {
    c.A = 1.f + t2 * (-1.f / 6.f + t2 * (1.f / 120.f + t2 * (-1.f / 5040.f)));
    c.B = 0.5f + t2 * (-1.f / 24.f + t2 * (1.f / 720.f + t2 * (-1.f / 40320.f)));
    c.C = 1.f / 6.f + t2 * (-1.f / 120.f + t2 * (1.f / 5040.f + t2 * (-1.f / 362880.f)));
    c.Ab = -1.f / 3.f + t2 * (1.f / 30.f + t2 * (-1.f / 840.f + t2 * (1.f / 45360.f)));
    c.Bb = -1.f / 12.f + t2 * (1.f / 180.f + t2 * (-1.f / 6720.f + t2 * (1.f / 453600.f)));
    c.Cb = -1.f / 60.f + t2 * (1.f / 1260.f + t2 * (-1.f / 60480.f + t2 * (1.f / 4989600.f)));
  } else {
    const float t = sqrtf(t2);
    float s, co;
    sincosf(t, &s, &co);
    const float sh = sinf(0.5f * t);
    c.A = s / t;
    c.B = 2.f * sh * sh / t2;
    c.C = (t - s) / (t2 * t);
    c.Ab = c.C - c.B;
    c.Bb = (c.A - 2.f * c.B) / t2;
    c.Cb = (c.B - 3.f * c.C) / t2;
  }
  return c;
}

// x' = exp_se3([w; v]) x = x + A w*x + B w*(w*x) + v + B w*v + C w*(w*v)   (warping.py:330-344)
__device__ __forceinline__ V3 se3_apply(V3 w, V3 v, V3 x) {
  const Se3Coef c = se3_coef(dot(w, w));
  const V3 wx = cross(w, x), wv = cross(w, v);
  const V3 wwx = cross(w, wx), wwv = cross(w, wv);
  return x + c.A * wx + c.B * wwx + v + c.B * wv + c.C * wwv;
}

// VJP of se3_apply for upstream g = dL/dx':  dL/dw, dL/dv  (dL/dx is not needed: sample points
// carry no parameters).
__device__ __forceinline__ void se3_vjp(V3 w, V3 v, V3 x, V3 g, V3& dw, V3& dv) {
  const Se3Coef c = se3_coef(dot(w, w));
  const V3 gw = cross(g, w);            // g x w
  const V3 wgw = cross(w, cross(w, g)); // w x (w x g)
  dv = g + c.B * gw + c.C * wgw;        // V^T g
  const V3 wx = cross(w, x), wv = cross(w, v);
  const V3 wwx = cross(w, wx), wwv = cross(w, wv);
  const float wg = dot(w, g);
  auto D = [&](V3 y) { return wg * y + dot(w, y) * g - 2.f * dot(y, g) * w; };
  const float sa = c.Ab * dot(g, wx) + c.Bb * (dot(g, wwx) + dot(g, wv)) + c.Cb * dot(g, wwv);
  dw = c.A * cross(x, g) + c.B * (D(x) + cross(v, g)) + c.C * D(v) + sa * w;
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
template <bool STASH>
__global__ __launch_bounds__(256, 2) void se3_warp_fwd_kernel(const WarpFwdArgs A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* act = smem;                  // [128][64] swizzled
  float* win = smem + WACT_FLOATS;    // [PKw][64] trunk input; reused as scratch after the skip layer
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane, part = wave;    // per-row phases: 4 threads per tile row
  const float* __restrict__ prm = A.params;
  const int PKw = A.PKw;
  const int PKS = (PKw + 31) / 32 * 32;
  const float4* wpk4 = reinterpret_cast<const float4*>(A.wpk);
  const size_t st_layer = (size_t)A.ntiles * FRAG_TILE_128;

  for (int tile = blockIdx.x; tile < A.ntiles; tile += gridDim.x) {
    // ---- prologue: sample point, AnnealedSinusoidalEncoder (modules.py:231-294), GLO code ----
    float x[3];
    const int row = tile * TILE_ROWS + p;
    {
      const int r = row < A.rows ? row : A.rows - 1;
      int id;
      if (A.points_in) {
        x[0] = A.points_in[3 * r]; x[1] = A.points_in[3 * r + 1]; x[2] = A.points_in[3 * r + 2];
        id = A.point_ids[r];
      } else {
        const int ray = r / A.S;
        const float z = A.zvals[r];
#pragma unroll
        for (int c = 0; c < 3; ++c)   // origins + z_vals * directions  (model_utils.py:72-73)
          x[c] = __fadd_rn(A.origins[3 * ray + c], __fmul_rn(z, A.directions[3 * ray + c]));
        id = A.warp_ids[ray];
      }
      float* stp = STASH ? A.st_win + (size_t)tile * PKS * TILE_ROWS : nullptr;
      auto put = [&](int k, float v) {
        win[k * TILE_ROWS + p] = v;
        if (STASH) stp[frag_index(k, p)] = v;
      };
      if (part == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) put(c, x[c]);
        if (A.points_raw && row < A.rows) {
          A.points_raw[3 * row] = x[0]; A.points_raw[3 * row + 1] = x[1]; A.points_raw[3 * row + 2] = x[2];
        }
      } else if (part == 1) {
        const float* __restrict__ code = prm + A.po.embed + (int64_t)id * A.G;   // glo.py:50-53
        for (int g = 0; g < A.G; ++g) put(3 + 6 * A.F + g, code[g]);
        for (int k = A.Win; k < PKw; ++k) put(k, 0.f);
        if (STASH) for (int k = PKw; k < PKS; ++k) stp[frag_index(k, p)] = 0.f;
      }
      const float half_pi = 1.57079632679489661923f;
      const float pi = 3.14159265358979323846f;
      for (int f = part; f < A.F; f += 4) {
        // cosine_easing_window (modules.py:274-294): 0.5 (1 + cos(pi clip(alpha - band, 0, 1) + pi))
        const float cl = fminf(fmaxf(A.alpha - (float)f, 0.f), 1.f);
        const float wdw = 0.5f * (1.f + cosf(__fadd_rn(__fmul_rn(pi, cl), pi)));
        const float fr = (float)(1 << f);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float a = __fmul_rn(x[c], fr);
          put(3 + (2 * f) * 3 + c, wdw * sinf(a));
          put(3 + (2 * f + 1) * 3 + c, wdw * sinf(__fadd_rn(a, half_pi)));
        }
      }
    }
    __syncthreads();

    // ---- trunk: 6 x Dense(128)+ReLU, skip concat [h, inputs] at layer 4 (warping.py:264-269) ----
    f32x16 acc[2][1];
    const int nq_in = PKw / 16, nit_in = PKw / 8;
    const float4* wL0 = wpk4 + (A.pk.fwd_L[0] / 4) + wave * nit_in * 64;
    WQuad<1> wnext = prefetch_quad<1>(wL0, lane);
#pragma unroll 1
    for (int l = 0; l < WARP_DEPTH; ++l) {
      bias_acc<1>(acc, prm + A.po.trunk_b[l], wave * 32, lane);
      if (l == 0) {
        mfma_k_loop<1, false>(acc, win, nq_in, wL0, lane, wnext);
      } else {
        mfma_k_loop<1, true>(acc, act, 8, wpk4 + (A.pk.fwd_L[l] / 4) + wave * 16 * 64, lane, wnext);
        if (l == WARP_SKIP) {
          const float4* w4b = wpk4 + (A.pk.fwd_L4b / 4) + wave * nit_in * 64;
          mfma_k_loop<1, false>(acc, win, nq_in, w4b, lane, prefetch_quad<1>(w4b, lane));
        }
      }
      wnext = prefetch_quad<1>(wpk4 + (A.pk.fwd_L[l + 1 < WARP_DEPTH ? l + 1 : l] / 4) + wave * 16 * 64, lane);
      __builtin_amdgcn_sched_barrier(0);
      fwd_epilogue<1, true, STASH>(
          acc, wave * 32, act,
          make_rsrc(STASH ? A.st_h + l * st_layer + (size_t)tile * FRAG_TILE_128 : nullptr, FRAG_TILE_128 * 4),
          wave * 8 * 1024, STASH ? A.bits + (((size_t)l * A.ntiles + tile) * 4 + wave) * 64 : nullptr, lane);
    }

    // ---- heads: w = Dense(128->3)(h), v = Dense(128->3)(h)  (warping.py:271-288, 328-329) ----
    {
      const float* __restrict__ ww = prm + A.po.w_k;   // [128][3]
      const float* __restrict__ wv = prm + A.po.v_k;
      float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const int k0 = part * 32;
      for (int k = k0; k < k0 + 32; ++k) {
        const float a = act[act_elem(k, p)];
#pragma unroll
        for (int c = 0; c < 3; ++c) { s[c] = fmaf(a, ww[3 * k + c], s[c]); s[3 + c] = fmaf(a, wv[3 * k + c], s[3 + c]); }
      }
#pragma unroll
      for (int c = 0; c < 6; ++c) win[(6 * part + c) * TILE_ROWS + p] = s[c];
      __syncthreads();
      if (part == 0) {
#pragma unroll
        for (int c = 0; c < 6; ++c)
          s[c] = (win[c * TILE_ROWS + p] + win[(6 + c) * TILE_ROWS + p]) + (win[(12 + c) * TILE_ROWS + p] + win[(18 + c) * TILE_ROWS + p]) +
                 prm[(c < 3 ? A.po.w_b + c : A.po.v_b + c - 3)];
        const V3 xw = se3_apply(v3(s[0], s[1], s[2]), v3(s[3], s[4], s[5]), v3(x[0], x[1], x[2]));
        float* o = A.points_out + (size_t)row * 3;
        o[0] = xw.x; o[1] = xw.y; o[2] = xw.z;
        if (STASH) {
          A.st_wv[2 * (size_t)row] = make_float4(s[0], s[1], s[2], 0.f);
          A.st_wv[2 * (size_t)row + 1] = make_float4(s[3], s[4], s[5], 0.f);
        }
      }
      __syncthreads();   // scratch (aliases win) is free again for the next tile's prologue
    }
  }
}

void launch_warp_fwd(const WarpFwdArgs& a, bool stash, int grid, hipStream_t stream) {
  const int pk = a.PKw < 32 ? 32 : a.PKw;   // the head scratch needs 24 rows
  const size_t lds = (size_t)(WACT_FLOATS + pk * TILE_ROWS) * sizeof(float);
  if (stash) {
    (void)hipFuncSetAttribute((const void*)se3_warp_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(se3_warp_fwd_kernel<true>, dim3(grid), dim3(256), lds, stream, a);
  } else {
    (void)hipFuncSetAttribute((const void*)se3_warp_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(se3_warp_fwd_kernel<false>, dim3(grid), dim3(256), lds, stream, a);
  }
}

// ---------------------------------------------------------------------------------------------
// backward (data gradients of the trunk, bias gradients, GLO-code gradient)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void se3_warp_bwd_kernel(const WarpBwdArgs A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* act = smem;                       // [128][64] swizzled: current dpre tile
  float* dwv = smem + WACT_FLOATS;         // [8][64]: dL/dw (0..2), dL/dv (3..5) of the tile rows
  float* dcs = dwv + 8 * TILE_ROWS;        // [8][64]: dL/dcode of the tile rows
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int p = lane, part = wave;
  const float* __restrict__ prm = A.params;
  const float4* wpk4 = reinterpret_cast<const float4*>(A.wpk);
  const size_t layer_fl = (size_t)A.ntiles * FRAG_TILE_128;
  const int PKS = (A.PKw + 31) / 32 * 32;
  const int n = wave * 32 + j;             // this lane's trunk column

  float db[WARP_DEPTH];
#pragma unroll
  for (int l = 0; l < WARP_DEPTH; ++l) db[l] = 0.f;
  float hsum[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // threads < 64: column sums of (dw, dv)

#pragma unroll 1
  for (int tile = blockIdx.x; tile < A.ntiles; tile += gridDim.x) {
    // ---- exp_se3 VJP per row ----
    if (tid < TILE_ROWS) {
      const int row = tile * TILE_ROWS + tid;
      V3 dw = v3(0.f, 0.f, 0.f), dv = dw;
      if (row < A.rows) {
        const float* sw = A.st_win + (size_t)tile * PKS * TILE_ROWS;
        const V3 x = v3(sw[frag_index(0, tid)], sw[frag_index(1, tid)], sw[frag_index(2, tid)]);
        const float4 w4 = A.st_wv[2 * (size_t)row], v4 = A.st_wv[2 * (size_t)row + 1];
        const V3 g = v3(A.d_points[3 * (size_t)row], A.d_points[3 * (size_t)row + 1], A.d_points[3 * (size_t)row + 2]);
        se3_vjp(v3(w4.x, w4.y, w4.z), v3(v4.x, v4.y, v4.z), x, g, dw, dv);
      }
      dwv[tid] = dw.x; dwv[TILE_ROWS + tid] = dw.y; dwv[2 * TILE_ROWS + tid] = dw.z;
      dwv[3 * TILE_ROWS + tid] = dv.x; dwv[4 * TILE_ROWS + tid] = dv.y; dwv[5 * TILE_ROWS + tid] = dv.z;
      A.d_w4[row] = make_float4(dw.x, dw.y, dw.z, 0.f);
      A.d_v4[row] = make_float4(dv.x, dv.y, dv.z, 0.f);
      hsum[0] += dw.x; hsum[1] += dw.y; hsum[2] += dw.z; hsum[3] += dv.x; hsum[4] += dv.y; hsum[5] += dv.z;
    }
    __syncthreads();

    // ---- heads^T (6 -> 128) on the VALU, ReLU mask of trunk layer 5 -> dpre_5 ----
    {
      float wh[6];
#pragma unroll
      for (int c = 0; c < 3; ++c) { wh[c] = prm[A.po.w_k + 3 * n + c]; wh[3 + c] = prm[A.po.v_k + 3 * n + c]; }
      const uint32_t mb = A.bits[(((size_t)(WARP_DEPTH - 1) * A.ntiles + tile) * 4 + wave) * 64 + lane];
      const __amdgpu_buffer_rsrc_t dy =
          make_rsrc(A.dy + (size_t)(WARP_DEPTH - 1) * layer_fl + (size_t)tile * FRAG_TILE_128, FRAG_TILE_128 * 4);
      float bsum = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int g = q_granule(q, h);
        float4 v4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          const float4 d = *reinterpret_cast<const float4*>(dwv + c * TILE_ROWS + 4 * g);
          v4.x = fmaf(d.x, wh[c], v4.x); v4.y = fmaf(d.y, wh[c], v4.y);
          v4.z = fmaf(d.z, wh[c], v4.z); v4.w = fmaf(d.w, wh[c], v4.w);
        }
        v4 = mask4(v4, (mb >> (4 * q)) & 15u);
        bsum += (v4.x + v4.y) + (v4.z + v4.w);
        *reinterpret_cast<float4*>(act + act_addr(n, g)) = v4;
        buf_store4(v4, dy, lane * 16, (wave * 8 + q) * 1024);
      }
      db[WARP_DEPTH - 1] += bsum;
    }
    __syncthreads();

    // GLO-code gradient: d code[g] = dpre_l . W_l[row_base + g][:]^T for the two layers that see the
    // input (l = 4 via the skip rows, l = 0), K = 128 on the VALU; thread = (row p, codes 2*part, 2*part+1).
    float dcode[2] = {0.f, 0.f};
    auto code_grad = [&](int64_t krow_off) {
      const int g0 = 2 * part;
      for (int k = 0; k < WARP_W; ++k) {
        const float a = act[act_elem(k, p)];
#pragma unroll
        for (int q = 0; q < 2; ++q)
          if (g0 + q < A.G) dcode[q] = fmaf(a, prm[krow_off + (int64_t)(g0 + q) * WARP_W + k], dcode[q]);
      }
    };

    // ---- l = 5..1: d h_l = dpre_l . W_l[0:128]^T ; mask h_l > 0 -> dpre_{l-1} ----
    f32x16 acc[2][1];
    WQuad<1> wnext = prefetch_quad<1>(wpk4 + (A.pk.bwd_LT[WARP_DEPTH - 1] / 4) + wave * 16 * 64, lane);
#pragma unroll 1
    for (int l = WARP_DEPTH - 1; l >= 1; --l) {
      if (l == WARP_SKIP) code_grad(A.po.trunk_k[WARP_SKIP] + (int64_t)(WARP_W + 3 + 6 * A.F) * WARP_W);
      const uint32_t mb = A.bits[(((size_t)(l - 1) * A.ntiles + tile) * 4 + wave) * 64 + lane];
      zero_acc<1>(acc);
      mfma_k_loop<1, true>(acc, act, 8, wpk4 + (A.pk.bwd_LT[l] / 4) + wave * 16 * 64, lane, wnext);
      wnext = prefetch_quad<1>(wpk4 + (A.pk.bwd_LT[l > 1 ? l - 1 : 1] / 4) + wave * 16 * 64, lane);
      __builtin_amdgcn_sched_barrier(0);
      const __amdgpu_buffer_rsrc_t dy = make_rsrc(A.dy + (size_t)(l - 1) * layer_fl + (size_t)tile * FRAG_TILE_128, FRAG_TILE_128 * 4);
      __syncthreads();
      float bsum = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        float4 v = acc_piece<1>(acc, 0, q);
        v = mask4(v, (mb >> (4 * q)) & 15u);
        bsum += (v.x + v.y) + (v.z + v.w);
        *reinterpret_cast<float4*>(act + act_addr(n, q_granule(q, h))) = v;
        buf_store4(v, dy, lane * 16, (wave * 8 + q) * 1024);
      }
#pragma unroll
      for (int q = 0; q < WARP_DEPTH; ++q)
        if (q == l - 1) db[q] += bsum;
      __syncthreads();
    }
    code_grad(A.po.trunk_k[0] + (int64_t)(3 + 6 * A.F) * WARP_W);

    // ---- per-ray sums of d code -> scatter-add into the embedding-table gradient ----
#pragma unroll
    for (int q = 0; q < 2; ++q) dcs[(2 * part + q) * TILE_ROWS + p] = dcode[q];
    __syncthreads();
    if (tid < 64) {
      const int g = tid & 7, sect = tid >> 3;      // 8 codes x 8 sections of 8 rows
      if (g < A.G) {
        const int row0 = 8 * sect;
        const int nvalid = A.rows - tile * TILE_ROWS;
        float s = 0.f;
        int cur = -1;
        for (int q = row0; q < row0 + 8 && q < nvalid; ++q) {
          const int grow = tile * TILE_ROWS + q;
          const int id = A.point_ids ? A.point_ids[grow] : A.warp_ids[grow / A.S];
          if (id != cur) {
            if (cur >= 0 && s != 0.f) atomicAdd(A.grad_embed + (size_t)cur * A.G + g, s);
            s = 0.f; cur = id;
          }
          s += dcs[g * TILE_ROWS + q];
        }
        if (cur >= 0 && s != 0.f) atomicAdd(A.grad_embed + (size_t)cur * A.G + g, s);
      }
    }
    __syncthreads();
  }

  // ---- flush the per-workgroup bias partials ----
  float* sp = A.small_part + (size_t)blockIdx.x * WARP_SMALL_PART;
#pragma unroll
  for (int l = 0; l < WARP_DEPTH; ++l) {
    const float v = db[l] + __shfl_xor(db[l], 32);
    if (h == 0) sp[WSP_DB_TRUNK + l * WARP_W + n] = v;
  }
  __syncthreads();
  if (tid < TILE_ROWS) {
#pragma unroll
    for (int c = 0; c < 6; ++c) dwv[c * TILE_ROWS + tid] = hsum[c];
  }
  __syncthreads();
  if (tid < 6) {
    float s = 0.f;
    for (int q = 0; q < TILE_ROWS; ++q) s += dwv[tid * TILE_ROWS + q];
    sp[WSP_DB_W + tid] = s;
  }
}

void launch_warp_bwd(const WarpBwdArgs& a, int grid, hipStream_t stream) {
  const size_t lds = (size_t)(WACT_FLOATS + 16 * TILE_ROWS) * sizeof(float);
  (void)hipFuncSetAttribute((const void*)se3_warp_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(se3_warp_bwd_kernel, dim3(grid), dim3(256), lds, stream, a);
}

}  // namespace nrf
