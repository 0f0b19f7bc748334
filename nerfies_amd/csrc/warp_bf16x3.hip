// SE(3) warp field with a split-bf16 ("bf16x3", float32-emulating) trunk: the inference forward of NRF_FLAG_BF16X3 when the model
// warps (what eval.py renders).  Arithmetic and design: mlp_bf16x3.hip / bf16x3_chain.h (every float32 operand a bf16 pair hi + lo, a
// product as hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_bf16, float32 accumulate); layer structure: warp_bf16.hip (a wave owns 32 rows and all 128 features, a layer is two panels of two blocks).
//
// Replaces, in that mode (reference /root/reference/nerfies):
//   modules.AnnealedSinusoidalEncoder   modules.py:231-294   prologue, float32 (sinf / cosf as warp_chain.hip), split into (hi, lo)
//   glo.GloEncoder                      glo.py:22-53         per-row code gather in the prologue
//   warping.SE3Field.warp               warping.py:322-353   trunk in split-bf16; heads' outputs (w, v) are float32 accumulators
//   rigid_body.exp_se3                  rigid_body.py:54-89  float32 closed form per row (se3_math.h)
// TranslationField (warping.py:62-199) and the warp_kwargs trunk shapes arrive as the padded internal image (nrf_plan.hip).
// No stash, no tangent pass: training and the Jacobian output keep the float32 kernels (warp_chain.hip).
#include "bf16x3_chain.h"
#include "se3_math.h"

namespace nrf {

namespace {

// Waves per workgroup.  The 128-wide trunk needs 2 x 2 x 32 packed registers + 64 accumulators: it fits 256 VGPRs (4 spilled outside the
// chains), so -- unlike the 256-wide NeRF chain -- two waves share a SIMD and one wave's epilogue VALU runs under the other's MFMAs.
constexpr int SE3_NW = 8;
// chunks of the stream in execution order (KiB): L0 2 x 18 | L1..L3 2 x 34 each | L4 (skip) [34 | 16] x 2 | L5 2 x 34 | heads 17
constexpr int XF_L0 = 18 * BF_KB, XF_T = 34 * BF_KB, XF_C = 16 * BF_KB, XF_HD = 17 * BF_KB;
constexpr int XF_TOTAL = 2 * XF_L0 + 4 * 2 * XF_T + 2 * (XF_T + XF_C) + XF_HD;
static_assert(XF_TOTAL == BFW_X3_STREAM_KB * BF_KB, "SE3 x3 stream length (nrf_internal.h)");

// One 128 -> 128 layer: 2 panels of 2 blocks, each ONE chunk (bias + 8 k-steps: 50 MFMAs) and, on the skip layer, a second chunk (the
// trunk input's 4 k-steps).  in = (ihi, ilo); its blocks 2, 3 arrive from acc1 = the previous layer's second panel during the first
// 24 slots of panel 0 (k-steps 4..7 start at slot 26); out blocks 0, 1 are written, 2, 3 stay pending in acc1.
// b2_*: the chunk two behind each of the layer's chunks (bf16_chain.h).
template <bool SKIP>
__device__ __forceinline__ void layer128_x3(ChainCtx& c, f32x16 (&acc0)[2], f32x16 (&acc1)[2], unsigned (&ihi)[4][8], unsigned (&ilo)[4][8],
                                            unsigned (&ohi)[4][8], unsigned (&olo)[4][8], const unsigned (&whi)[2][8],
                                            const unsigned (&wlo)[2][8], int b2_p0, int b2_p0c, int b2_p1, int b2_p1c, const bf16x8 bias_op) {
  auto bA = [&](int t, bool lo) __attribute__((always_inline)) { return lo ? kop(ilo, t) : kop(ihi, t); };
  auto bC = [&](int t, bool lo) __attribute__((always_inline)) { return lo ? kop(wlo, t) : kop(whi, t); };
  auto none = [&](int) __attribute__((always_inline)) {};
  X3_PANEL_NW(SE3_NW, 2, true, 8, true, 24, x3_ops(true), acc0, b2_p0, bA, [&](int k) __attribute__((always_inline)) { x3_epi<24, 2, true>(k, acc1, ihi, ilo); });
  if constexpr (SKIP) X3_PANEL_NW(SE3_NW, 2, false, 4, false, 0, 0, acc0, b2_p0c, bC, none);
  X3_PANEL_NW(SE3_NW, 2, true, 8, true, 49, x3_ops(true), acc1, b2_p1, bA, [&](int k) __attribute__((always_inline)) { x3_epi<49, 0, true>(k, acc0, ohi, olo); });
  if constexpr (SKIP) X3_PANEL_NW(SE3_NW, 2, false, 4, false, 0, 0, acc1, b2_p1c, bC, none);
}

}  // namespace

// One workgroup (SE3_NW waves) per CU; 32 SE3_NW rows per workgroup iteration, one 32-row group per wave.
__global__ __launch_bounds__(64 * SE3_NW) __attribute__((amdgpu_waves_per_eu(SE3_NW / 4, SE3_NW / 4))) void se3_fwd_x3_kernel(const WarpFwdArgs A) {
  extern __shared__ __attribute__((aligned(16))) char bf_lds[];
  const int lane0 = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int niter = (A.rows + 32 * SE3_NW - 1) / (32 * SE3_NW);
  const bf16x8 bias_op = as_bf16x8(0x3F803F80u, 0x00003F80u, 0u, 0u);   // B = 1 in k-slots 0, 1, 2 (bias hi + lo + lo2)

  // cosine_easing_window (modules.py:274-294) of every band, once per kernel, in SGPRs: 0.5 (1 + cos(pi clip(alpha - f, 0, 1) + pi))
  float wnd[10];
  {
    const float warp_alpha = A.dyn ? A.dyn->warp_alpha : A.alpha;
    const float pi = 3.14159265358979323846f;
#pragma unroll
    for (int f = 0; f < 10; ++f) {
      const float cl = fminf(fmaxf(warp_alpha - (float)f, 0.f), 1.f);
      wnd[f] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(0.5f * (1.f + cosf(__fadd_rn(__fmul_rn(pi, cl), pi))))));
    }
  }

  ChainCtx c;
  chain_start<X3_SLOT>(c, bf_lds, A.bwpk, XF_TOTAL, XF_L0, XF_L0, lane0, wave, SE3_NW);

#pragma unroll 1
  for (int it = blockIdx.x; it < niter; it += gridDim.x) {
    int lo_ = lane0;
    asm volatile("" : "+v"(lo_));   // per-iteration opaque lane (mlp_bf16.hip)
    const int lane = lo_, n = lane & 31, h = lane >> 5;
    const int row = (it * SE3_NW + wave) * 32 + n;
    const int rc = row < A.rows ? row : A.rows - 1;
    const int F = A.F, cbase = 3 + 6 * F;   // first code feature
    // ---- the point, its warp id ----
    float x[3];
    int id;
    if (A.points_in) {
      x[0] = A.points_in[3 * rc]; x[1] = A.points_in[3 * rc + 1]; x[2] = A.points_in[3 * rc + 2];
      id = A.point_ids[rc];
    } else {
      const int ray = rc / A.S;
      const float z = A.zvals[rc];
#pragma unroll
      for (int k = 0; k < 3; ++k) x[k] = __fadd_rn(A.origins[3 * ray + k], __fmul_rn(z, A.directions[3 * ray + k]));   // model_utils.py:72-73
      id = A.warp_ids ? A.warp_ids[ray] : ray;   // nullptr: per-ray codes (metadata_encoded / TimeEncoder output)
    }
    if (A.points_raw && h == 0 && row < A.rows) {
      A.points_raw[3 * (size_t)row] = x[0]; A.points_raw[3 * (size_t)row + 1] = x[1]; A.points_raw[3 * (size_t)row + 2] = x[2];
    }
    const float* __restrict__ code = A.embed_table + (int64_t)id * A.G;   // glo.py:50-53
    // ---- trunk input [annealed posenc(x), code] (warping.py:326-327), float32 as warp_chain.hip, split into the
    //      (hi, lo) B-operand registers: lane (n, h) holds features 32 b + 8 j + 4 h + i ----
    unsigned whi[2][8], wlo[2][8];
    {
      const float half_pi = 1.57079632679489661923f;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          float v[2];
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int r16 = 2 * q + t;
            const int e = 32 * b + 8 * (r16 >> 2) + 4 * h + (r16 & 3);
            float val = 0.f;
            if (e < 3) {
              val = e == 0 ? x[0] : e == 1 ? x[1] : x[2];
            } else if (e < cbase) {
              const int idx = e - 3, f = idx / 6, rem = idx - 6 * f, cc = rem >= 3 ? rem - 3 : rem;
              const float a = __fmul_rn(cc == 0 ? x[0] : cc == 1 ? x[1] : x[2], (float)(1 << f));
              float wdw = wnd[0];
#pragma unroll
              for (int j = 1; j < 10; ++j) wdw = f == j ? wnd[j] : wdw;
              val = wdw * sinf(rem >= 3 ? __fadd_rn(a, half_pi) : a);
            } else if (e < cbase + A.G) {
              val = code[e - cbase];
            }
            v[t] = val;
          }
          const unsigned ph = pack_bf16(v[0], v[1]);
          whi[b][q] = ph;
          wlo[b][q] = pack_bf16(v[0] - __uint_as_float(ph << 16), v[1] - __uint_as_float(ph & 0xFFFF0000u));
        }
    }

    unsigned ua[4][8], ual[4][8], ub[4][8], ubl[4][8];
    f32x16 acc0[2], acc1[2];
    auto none = [&](int) __attribute__((always_inline)) {};
    // ---- L0: trunk input -> ua; 2 panels x one chunk of bias + 4 k-steps ----
    {
      auto b0 = [&](int t, bool lo) __attribute__((always_inline)) { return lo ? kop(wlo, t) : kop(whi, t); };
      X3_PANEL_NW(SE3_NW, 2, true, 4, true, 0, 0, acc0, XF_T, b0, none);
      X3_PANEL_NW(SE3_NW, 2, true, 4, true, 25, x3_ops(true), acc1, XF_T, b0, [&](int k) __attribute__((always_inline)) { x3_epi<25, 0, true>(k, acc0, ua, ual); });
    }
    // ---- trunk: 6 x Dense(128) + ReLU, skip concat [h, inputs] at layer 4 (warping.py:264-269) ----
    layer128_x3<false>(c, acc0, acc1, ua, ual, ub, ubl, whi, wlo, XF_T, 0, XF_T, 0, bias_op);              // L1: ua -> ub
    layer128_x3<false>(c, acc0, acc1, ub, ubl, ua, ual, whi, wlo, XF_T, 0, XF_T, 0, bias_op);              // L2: ub -> ua
    layer128_x3<false>(c, acc0, acc1, ua, ual, ub, ubl, whi, wlo, XF_T, 0, XF_C, 0, bias_op);              // L3: ua -> ub (two behind: L4's chunks)
    layer128_x3<true>(c, acc0, acc1, ub, ubl, ua, ual, whi, wlo, XF_T, XF_C, XF_T, XF_T, bias_op);          // L4 (skip): ub -> ua
    layer128_x3<false>(c, acc0, acc1, ua, ual, ub, ubl, whi, wlo, XF_HD, 0, XF_L0, 0, bias_op);            // L5: ua -> ub
    // ---- heads: w = Dense(128 -> 3)(h6), v = Dense(128 -> 3)(h6) (warping.py:271-288, 328-329): one block, features 0..2 = w,
    //      3..5 = v; h6's blocks 2, 3 (pending in acc1) are this chunk's k-steps 4..7 = slots 13.. ----
    f32x16 hd[1];
    {
      auto bh = [&](int t, bool lo) __attribute__((always_inline)) { return lo ? kop(ubl, t) : kop(ub, t); };
      X3_PANEL_NW(SE3_NW, 1, true, 8, true, 12, x3_ops(true), hd, XF_L0, bh, [&](int k) __attribute__((always_inline)) { x3_epi<12, 2, true>(k, acc1, ub, ubl); });
    }
    // lane (n, 0): registers 0..3 = (w0, w1, w2, v0); lane (n, 1): registers 0, 1 = (v1, v2)
    const float v1 = __shfl_xor(hd[0][0], 32), v2 = __shfl_xor(hd[0][1], 32);
    if (h == 0 && row < A.rows) {
      const V3 w = v3(hd[0][0], hd[0][1], hd[0][2]), v = v3(hd[0][3], v1, v2);
      const V3 xw = se3_apply(w, v, v3(x[0], x[1], x[2]));
      float* o = A.points_out + (size_t)row * 3;
      o[0] = xw.x; o[1] = xw.y; o[2] = xw.z;
    }
  }
}

void launch_warp_fwd_x3(const WarpFwdArgs& a, int max_grid, hipStream_t stream) {
  (void)hipFuncSetAttribute((const void*)se3_fwd_x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)X3_LDS_BYTES);
  const int nit = (a.rows + 32 * SE3_NW - 1) / (32 * SE3_NW);
  hipLaunchKernelGGL(se3_fwd_x3_kernel, dim3(nit < max_grid ? nit : max_grid), dim3(64 * SE3_NW), X3_LDS_BYTES, stream, a);
}

}  // namespace nrf
