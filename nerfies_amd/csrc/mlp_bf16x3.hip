// Split-bf16 ("bf16x3") inference chain of the NeRF MLP: float32-EMULATING arithmetic on the bf16 matrix pipe (NRF_FLAG_BF16X3).
//
// Why.  The certified-parity path (mlp_chain.hip) is capped by v_mfma_f32_32x32x2_f32: 157 TFLOP/s, 1/16 of the bf16 pipe.  A float32
// x is hi + lo + O(2^-17 |x|) with hi = bf16(x), lo = bf16(x - hi), so
//     w x  =  w_hi x_hi + w_lo x_hi + w_hi x_lo  +  O(2^-16 |w x|)          (the lo . lo term and the two residuals dropped)
// -- three v_mfma_f32_32x32x16_bf16 per product, accumulated in float32, against eight fp32 MFMAs for the same 16 k values.
// Arithmetic: modules.py:26-62 (MLP), modules.py:95-169 (NerfMLP), models.py:270-277 (activations), evaluated to ~1e-6 of the
// float32 chains on rendered colour (tests/test_gpu_bf16x3.py against the unmodified reference's outputs); NOT bit-comparable with
// them and never reported as "f32": bench.py says dtype "bf16x3 (fp32-emulating)".
//
// Design = bf16_chain.h (transposed panel-outer chain, weights through a three-slot LDS ring, activations in registers), with
//   * every k-step row of the weight stream doubled -- (W_hi, W_lo) -- by the pack kernel (RcPackDesc.x3); the inner loop (x3_panel
//     below) reads a W_hi fragment ONCE for its two MFMAs (against x_hi and x_lo) and a W_lo fragment for one: 2/3 ds_read_b128 per
//     MFMA and 2/3 of the ring refill of a tripled stream.  A four-wave workgroup refills 1 KiB per 4 MFMAs -- twice the eight-wave
//     bf16 chains' rate -- and the LDS array was the busiest unit of the first version (tripled rows through bf_panel unchanged:
//     7.6 ms for the fine level of an 8192-ray chunk; W_hi once: 6.6 ms; same-box A/B, profiles/r06_experiments.md section 5);
//   * the refill of chunk g+2 spread one piece per MFMA behind the chunk's barrier instead of one burst (a wave is alone on its
//     SIMD: nobody else feeds the matrix pipe while it issues ~60 scalar / VMEM instructions): 6.6 -> 6.1 ms;
//   * a panel's rows cut into chunks of <= 17 rows (34 KiB ring slots, 102 of the CU's 160 KiB);
//   * activations kept as TWO packed register sets (hi, lo): 2 x 64 registers per 256-wide layer and side, i.e. a wave needs most of
//     the 512-register file (460-486 allocated, nothing in scratch): four waves per workgroup, one per SIMD, 128 samples per iteration;
//   * the epilogue unit (ReLU, hi = pack, lo = pack(x - hi): 9 VALU per register pair + the moves between the VGPR and AGPR halves)
//     rides between the MFMAs of the next panel;
//   * posenc by sinf (the float32 chains' function), bias as a (hi, lo, lo2) triple against B = 1.
// Not a lever here (same-box A/B): the fragment prefetch depth (8 vs 12), the VALU group sizes of the riding epilogue, laundering the
// slot bases so the ds_read offsets fit their immediates (that one LOSES 20 %).
// Inference only: the training path keeps its float32 / bf16 stashes.
#include <stdlib.h>

#include "bf16x3_chain.h"
#include "philox.h"

namespace nrf {

namespace {

// ---- weight stream: chunk sizes in execution order (nrf_plan.hip emits the same rows; a chunk is a row range of a panel) ----
//   L0      4 panels x (bias + 2 x 4 k-steps of the posenc)                                   4 x 18 KiB
//   L1..L7  4 panels x [A: bias + 2 x 8 k-steps | B: 2 x 8 k-steps]; the skip layer + [C: 2 x 4 posenc k-steps]
//   BN      as a trunk layer;  AL: the alpha head, one block x (bias + 2 x 16)                 33
//   RG      2 panels x 2 chunks of 2 x 8 k-steps (the bias rides in the fp32 per-ray term)     4 x 32
//   LG      one block x (bias + 2 x 8)                                                         17
constexpr int XW_L0 = 18 * BF_KB, XW_A = 34 * BF_KB, XW_B = 32 * BF_KB, XW_C = 16 * BF_KB, XW_AL = 33 * BF_KB, XW_RG = 32 * BF_KB,
              XW_LG = 17 * BF_KB;
constexpr int XW_TOTAL = 4 * XW_L0 + 7 * 4 * (XW_A + XW_B) + 4 * (XW_A + XW_B + XW_C) + XW_AL + 4 * XW_RG + XW_LG;
static_assert(XW_TOTAL == BF_X3_STREAM_KB * BF_KB, "x3 stream length (nrf_internal.h)");

// One 256 -> 256 layer: 4 panels of 2 blocks, each cut into chunk A (bias + k-steps 0..7: 50 MFMAs), chunk B (k-steps 8..15: 48) and, on the
// skip layer, chunk C (the posenc's 4 k-steps).  in = (ihi, ilo); its blocks 6, 7 arrive from acc1 = the previous layer's last
// panel during chunk A of panel 0 (PEND; chunk A reads blocks 0..3 only); out blocks 0..5 are written, 6, 7 stay pending in acc1.
// n0 / n1: sizes of the NEXT layer's first two chunks (every call names the chunk two ahead: bf16_chain.h).
template <bool PEND, bool PRELU, bool RELU, bool SKIP>
__device__ __forceinline__ void layer256_x3(ChainCtx& c, f32x16 (&acc0)[2], f32x16 (&acc1)[2], unsigned (&ihi)[8][8], unsigned (&ilo)[8][8],
                                            unsigned (&ohi)[8][8], unsigned (&olo)[8][8], const unsigned (&phi)[2][8],
                                            const unsigned (&plo)[2][8], int n0, int n1, const bf16x8 bias_op) {
  auto bA = [&](int t, bool lo) __attribute__((always_inline)) { return lo ? kop(ilo, t) : kop(ihi, t); };
  auto bB = [&](int t, bool lo) __attribute__((always_inline)) { return lo ? kop(ilo, 8 + t) : kop(ihi, 8 + t); };
  auto bC = [&](int t, bool lo) __attribute__((always_inline)) { return lo ? kop(plo, t) : kop(phi, t); };
  auto none = [&](int) __attribute__((always_inline)) {};
  constexpr int OPP = x3_ops(PRELU), OPN = x3_ops(RELU);
  constexpr int a2 = SKIP ? XW_C : XW_A;    // chunk two behind an A chunk (inside the layer)
  constexpr int b2 = SKIP ? XW_A : XW_B;    // ... behind a B chunk
  // panel 0 -> acc0
  if constexpr (PEND)
    X3_PANEL(2, true, 8, true, 49, OPP, acc0, a2, bA, [&](int k) __attribute__((always_inline)) { x3_epi<49, 6, PRELU>(k, acc1, ihi, ilo); });
  else
    X3_PANEL(2, true, 8, true, 0, 0, acc0, a2, bA, none);
  X3_PANEL(2, false, 8, false, 0, 0, acc0, b2, bB, none);
  if constexpr (SKIP) X3_PANEL(2, false, 4, false, 0, 0, acc0, XW_B, bC, none);
  // panel 1 -> acc1; panel 0's epilogue rides in its chunk A
  X3_PANEL(2, true, 8, true, 49, OPN, acc1, a2, bA, [&](int k) __attribute__((always_inline)) { x3_epi<49, 0, RELU>(k, acc0, ohi, olo); });
  X3_PANEL(2, false, 8, false, 0, 0, acc1, b2, bB, none);
  if constexpr (SKIP) X3_PANEL(2, false, 4, false, 0, 0, acc1, XW_B, bC, none);
  // panel 2 -> acc0
  X3_PANEL(2, true, 8, true, 49, OPN, acc0, a2, bA, [&](int k) __attribute__((always_inline)) { x3_epi<49, 2, RELU>(k, acc1, ohi, olo); });
  X3_PANEL(2, false, 8, false, 0, 0, acc0, b2, bB, none);
  if constexpr (SKIP) X3_PANEL(2, false, 4, false, 0, 0, acc0, XW_B, bC, none);
  // panel 3 -> acc1 (pending on exit); its last chunks name the next layer's first two
  X3_PANEL(2, true, 8, true, 49, OPN, acc1, SKIP ? XW_C : n0, bA, [&](int k) __attribute__((always_inline)) { x3_epi<49, 4, RELU>(k, acc0, ohi, olo); });
  X3_PANEL(2, false, 8, false, 0, 0, acc1, SKIP ? n0 : n1, bB, none);
  if constexpr (SKIP) X3_PANEL(2, false, 4, false, 0, 0, acc1, n1, bC, none);
}

// sigma activation (models.py:276-277), as mlp_bf16.hip bf_sigma
__device__ __forceinline__ float x3_sigma(float x, int kind) {
  if (kind != 1) return fmaxf(x, 0.f);
  const float e = __expf(-fabsf(x)), u = 1.f + e, d = u - 1.f;
  return fmaxf(x, 0.f) + (d == 0.f ? e : __logf(u) * __fdividef(e, d));
}

}  // namespace

// One workgroup (4 waves, one per SIMD, 512 registers each) per CU; 128 samples per workgroup iteration, one 32-sample group per wave.
template <bool ABN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void nerf_mlp_fwd_x3_kernel(const ChainFwdArgs A) {
  extern __shared__ __attribute__((aligned(16))) char bf_lds[];
  const int lane0 = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int niter = (A.rows + 127) / 128;
  const bf16x8 bias_op = as_bf16x8(0x3F803F80u, 0x00003F80u, 0u, 0u);   // B = 1 in k-slots 0, 1, 2 (bias hi + lo + lo2)

  ChainCtx c;
  chain_start<X3_SLOT>(c, bf_lds, A.wpk, XW_TOTAL, XW_L0, XW_L0, lane0, wave, 4);

#pragma unroll 1
  for (int it = blockIdx.x; it < niter; it += gridDim.x) {
    int lo_ = lane0;
    asm volatile("" : "+v"(lo_));   // per-iteration opaque lane (mlp_bf16.hip)
    const int lane = lo_, n = lane & 31, h = lane >> 5;
    const int row = (it * 4 + wave) * 32 + n;
    const int rc = row < A.rows ? row : A.rows - 1;
    float x[3];
    if (A.points) {
      x[0] = A.points[3 * rc]; x[1] = A.points[3 * rc + 1]; x[2] = A.points[3 * rc + 2];
    } else {
      const int ray = rc / A.S;
      const float z = A.zvals[rc];
#pragma unroll
      for (int k = 0; k < 3; ++k) x[k] = __fadd_rn(A.origins[3 * ray + k], __fmul_rn(z, A.directions[3 * ray + k]));   // model_utils.py:72-73
    }
    // SinusoidalEncoder (modules.py:213-228) in fp32 (sinf, as mlp_chain.hip), split straight into B-operand registers
    unsigned phi[2][8], plo[2][8];
    {
      const float half_pi = 1.57079632679489661923f;   // fp32(pi/2), modules.py:222
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          float v[2];
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int r16 = 2 * q + t;
            const int e = 32 * b + 8 * (r16 >> 2) + 4 * h + (r16 & 3);   // posenc feature
            float val = 0.f;
            if (e < 3) {
              val = e == 0 ? x[0] : e == 1 ? x[1] : x[2];
            } else if (e < A.P) {
              const int idx = e - 3, f = idx / 6, rem = idx - 6 * f, cc = rem >= 3 ? rem - 3 : rem;
              const float a = __fmul_rn(cc == 0 ? x[0] : cc == 1 ? x[1] : x[2], (float)(1 << f));
              val = sinf(rem >= 3 ? __fadd_rn(a, half_pi) : a);
            }
            v[t] = val;
          }
          const unsigned ph = pack_bf16(v[0], v[1]);
          phi[b][q] = ph;
          plo[b][q] = pack_bf16(v[0] - __uint_as_float(ph << 16), v[1] - __uint_as_float(ph & 0xFFFF0000u));
        }
    }

    unsigned ua[8][8], ual[8][8], ub[8][8], ubl[8][8];
    f32x16 acc0[2], acc1[2];
    auto none = [&](int) __attribute__((always_inline)) {};
    // ---- L0: posenc -> ua; 4 panels x one chunk of bias + 12 rows ----
    {
      auto b0 = [&](int t, bool lo) __attribute__((always_inline)) { return lo ? kop(plo, t) : kop(phi, t); };
      X3_PANEL(2, true, 4, true, 0, 0, acc0, XW_L0, b0, none);
      X3_PANEL(2, true, 4, true, 25, x3_ops(true), acc1, XW_L0, b0, [&](int k) __attribute__((always_inline)) { x3_epi<25, 0, true>(k, acc0, ua, ual); });
      X3_PANEL(2, true, 4, true, 25, x3_ops(true), acc0, XW_A, b0, [&](int k) __attribute__((always_inline)) { x3_epi<25, 2, true>(k, acc1, ua, ual); });
      X3_PANEL(2, true, 4, true, 25, x3_ops(true), acc1, XW_B, b0, [&](int k) __attribute__((always_inline)) { x3_epi<25, 4, true>(k, acc0, ua, ual); });
    }
    // ---- trunk layers 1..7 (ua -> ub -> ua -> ub, the skip layer between the two rounds: one copy of the code, run twice) ----
#pragma unroll 1
    for (int t = 0; t < 2; ++t) {
      layer256_x3<true, true, true, false>(c, acc0, acc1, ua, ual, ub, ubl, phi, plo, XW_A, XW_B, bias_op);
      layer256_x3<true, true, true, false>(c, acc0, acc1, ub, ubl, ua, ual, phi, plo, XW_A, XW_B, bias_op);
      layer256_x3<true, true, true, false>(c, acc0, acc1, ua, ual, ub, ubl, phi, plo, XW_A, XW_B, bias_op);
      if (t == 0) layer256_x3<true, true, true, true>(c, acc0, acc1, ub, ubl, ua, ual, phi, plo, XW_A, XW_B, bias_op);   // skip: [h, posenc]
    }
    // ---- bottleneck (linear): h8 = ub -> ua; its last chunks name the alpha chunk and the rgb branch ----
    layer256_x3<true, true, false, false>(c, acc0, acc1, ub, ubl, ua, ual, phi, plo, XW_AL, XW_RG, bias_op);
    // ---- alpha head: one block on h8 (use_alpha_condition, modules.py:152-157: on the bottleneck + the per-ray appearance term);
    //      the bottleneck's pending blocks 6, 7 (k-steps 12..15 = rows 37..48 under ABN) ride in slots 1..32 ----
    float alpha_raw;
    {
      f32x16 aa[1];
      auto bs = [&](int t, bool lo) __attribute__((always_inline)) { return ABN ? (lo ? kop(ual, t) : kop(ua, t)) : (lo ? kop(ubl, t) : kop(ub, t)); };
      X3_PANEL(1, true, 16, true, 32, x3_ops(false), aa, XW_RG, bs, [&](int k) __attribute__((always_inline)) { x3_epi<32, 6, false>(k, acc1, ua, ual); });
      alpha_raw = aa[0][0];
      if constexpr (ABN) alpha_raw += A.alpha_ct[min(rc / A.S, A.B - 1)];
    }
    // ---- rgb branch: hidden 256 -> 128 (+ the fp32 per-ray condition term incl. bias), ReLU ----
    unsigned rh[4][8], rhl[4][8];
    const float* ct = A.condterm + (size_t)min(rc / A.S, A.B - 1) * RGB_W + 4 * h;
    auto add_ct = [&](f32x16 (&acc)[2], int o0) __attribute__((always_inline)) {
#pragma unroll
      for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 c4 = *reinterpret_cast<const float4*>(ct + 32 * (o0 + o) + 8 * j);
          acc[o][4 * j] += c4.x; acc[o][4 * j + 1] += c4.y; acc[o][4 * j + 2] += c4.z; acc[o][4 * j + 3] += c4.w;
        }
    };
    {
      auto r0 = [&](int t, bool lo) __attribute__((always_inline)) { return lo ? kop(ual, t) : kop(ua, t); };
      auto r1 = [&](int t, bool lo) __attribute__((always_inline)) { return lo ? kop(ual, 8 + t) : kop(ua, 8 + t); };
      X3_PANEL(2, false, 8, true, 0, 0, acc0, XW_RG, r0, none);
      X3_PANEL(2, false, 8, false, 0, 0, acc0, XW_RG, r1, none);
      add_ct(acc0, 0);
      X3_PANEL(2, false, 8, true, 47, x3_ops(true), acc1, XW_LG, r0, [&](int k) __attribute__((always_inline)) { x3_epi<47, 0, true>(k, acc0, rh, rhl); });
      X3_PANEL(2, false, 8, false, 0, 0, acc1, XW_L0, r1, none);
      add_ct(acc1, 2);
    }
    // ---- rgb logits: 128 -> 3 (one block), sigmoid; the pending panel (rgb hidden blocks 2, 3 = k-steps 4..7 = rows 13..24)
    //      rides in slots 1..12 ----
    {
      f32x16 lg[1];
      auto bl = [&](int t, bool lo) __attribute__((always_inline)) { return lo ? kop(rhl, t) : kop(rh, t); };
      X3_PANEL(1, true, 8, true, 12, x3_ops(true), lg, XW_L0, bl, [&](int k) __attribute__((always_inline)) { x3_epi<12, 2, true>(k, acc1, rh, rhl); });
      if (h == 0 && row < A.rows) {
        float4 o;
        o.x = 1.f / (1.f + expf(-lg[0][0]));
        o.y = 1.f / (1.f + expf(-lg[0][1]));
        o.z = 1.f / (1.f + expf(-lg[0][2]));
        float araw = alpha_raw;
        if (A.noise_std > 0.f)   // model_utils.noise_regularize (model_utils.py:266-282)
          araw += A.noise_std * (A.noise ? A.noise[row]
                                         : philox_normal(A.dyn ? A.dyn->rng_seed : A.noise_seed, A.dyn ? A.dyn->rng_offset : A.noise_offset, A.noise_stream, (uint32_t)row));
        o.w = x3_sigma(araw, A.sigma_act);
        A.out4[row] = o;
      }
    }
  }
}

namespace {
template <bool ABN>
void launch_x3_variant(const ChainFwdArgs& a, int max_grid, hipStream_t stream) {
  (void)hipFuncSetAttribute((const void*)nerf_mlp_fwd_x3_kernel<ABN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)X3_LDS_BYTES);
  const int nit = (a.rows + 127) / 128;
  hipLaunchKernelGGL((nerf_mlp_fwd_x3_kernel<ABN>), dim3(nit < max_grid ? nit : max_grid), dim3(256), X3_LDS_BYTES, stream, a);
}
}  // namespace

void launch_chain_fwd_x3(const ChainFwdArgs& a, int max_grid, hipStream_t stream) {
  if (a.alpha_ct != nullptr) launch_x3_variant<true>(a, max_grid, stream);
  else launch_x3_variant<false>(a, max_grid, stream);
}

}  // namespace nrf
