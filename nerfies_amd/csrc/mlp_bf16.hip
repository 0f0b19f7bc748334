// bf16-operand chains of the NeRF MLP: forward (inference / rendering, and with the training stash) and the data-gradient
// pass (BASELINE config D "bf16 MLP with fp32 composite").
//
// Opt-in mode (NRF_FLAG_BF16): activations, gradients and weights are rounded to bfloat16 (RNE) as MFMA operands; accumulation,
// biases (hi + lo bf16 pair), the per-ray condition term, the activations' ReLU and everything outside the MLP
// (sampling, compositing, loss, master weights, Adam) stay fp32.  Not bit-comparable with the fp32 path: tests bound it at
// ~1e-2 on rendered colour.  Matches modules.py:95-169 (NerfMLP) / modules.py:26-62 (MLP) with models.py:270-277's activations.
//
// Design: bf16_chain.h (panel-outer transposed chain, activations never leave the registers, weights streamed through a
// three-slot LDS ring by LDS-DMA, the epilogue of a panel issued between the MFMAs of the next one).
//
// Training (STASH): every layer's packed output registers -- which ARE the next layer's B operand -- are stored as they lie
// (nrf_internal.h BfStash: 1 KiB coalesced per wave store, non-temporal), plus one ReLU-derivative bit per pre-activation.
// The dgrad kernel below runs the same chain backwards,
//   dX^T[in feature][sample] = W . dY^T,   A = W as it is stored (K = the layer's output features),
// masks with the bits and stores each dpre in the same layout; wgrad_bf16.hip turns the two stashes into weight gradients.
#include <stdlib.h>

#include "bf16_chain.h"
#include "philox.h"

namespace nrf {

namespace {

// ---- forward weight stream: chunk sizes in execution order (nrf_plan.hip build_plan emits the same sequence) ----
//   L0      4 panels x (bias + 4 k-steps of the posenc)                        4 x 10 KiB
//   L1..L7  4 panels x (bias + 16 k-steps); skip layer + 4 posenc k-steps      4 x 34 KiB (4 x 42)
//   BN      4 panels x (bias + 16), then the alpha head as a one-block panel   4 x 34 + 17
//   RG      2 panels x 16 k-steps (the bias rides in the fp32 per-ray term)    2 x 32
//   LG      one block x (bias + 8 k-steps)                                     9
constexpr int FW_L0 = 10 * BF_KB, FW_T = 34 * BF_KB, FW_S = 42 * BF_KB, FW_AL = 17 * BF_KB, FW_RG = 32 * BF_KB, FW_LG = 9 * BF_KB;
constexpr int FW_TOTAL = 4 * FW_L0 + 28 * FW_T + 4 * FW_S + FW_AL + 2 * FW_RG + FW_LG;
static_assert(FW_TOTAL == BF_FWD_STREAM_KB * BF_KB, "forward stream length (nrf_internal.h)");

// VALU instructions per epilogue unit: pack, [ReLU], [sign bit: min + mad]
__device__ __forceinline__ constexpr int epi_ops(bool relu, bool stash) { return 1 + (relu ? 1 : 0) + (relu && stash ? 2 : 0); }

// Units of a pending panel (2 blocks = 16 packed registers) that fall on slot k: accumulators -> (ReLU) -> bf16 pairs in
// out[O0], out[O0 + 1]; training: sign bits into mb, 1 KiB stash store per half block (rs = the PANEL's 4 KiB of the stash)
template <int SPAN, int O0, bool RELU, bool STASH, int NBLK>
__device__ __forceinline__ void panel_epi(int k, const f32x16 (&pend)[2], unsigned (&out)[NBLK][8], unsigned& mb,
                                          __amdgpu_buffer_rsrc_t rs, int lane16) {
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    if (epi_slot(u, 16, SPAN) != k) continue;
    const int o = u >> 3, q = u & 7;
    unsigned pk = pack_bf16(pend[o][2 * q], pend[o][2 * q + 1]);
    if (RELU) pk = relu_pk(pk);
    out[O0 + o][q] = pk;
    if (STASH) {
      if (RELU) mb = bits_push(mb, pk);
      if ((q & 3) == 3) {
        const int jp = q >> 2;
        bf_store16(rs, lane16 + (o * 2 + jp) * BF_KB, out[O0 + o][4 * jp], out[O0 + o][4 * jp + 1], out[O0 + o][4 * jp + 2],
                   out[O0 + o][4 * jp + 3]);
      }
    }
  }
}

// One 256-wide layer of the forward chain, 4 panels of 2 blocks.  `in` = packed input (its blocks 6, 7 arrive from acc1 = the
// previous layer's last panel during chunk 0 when PEND), out = packed output blocks 0..5; blocks 6, 7 stay pending in acc1.
//   R: rows per panel (bias + k-steps); bsel(r): B operand of row r; PRELU / RELU: activation of the previous / this layer;
//   st_prev / st: their stash groups (8 blocks each); mbp / mbn: their sign-bit words; b2_*: sizes of the chunks two ahead
template <int R, bool PEND, bool PRELU, bool RELU, bool STASH, class BSel>
__device__ __forceinline__ void layer256(ChainCtx& c, f32x16 (&acc0)[2], f32x16 (&acc1)[2], unsigned (&in)[8][8], unsigned (&out)[8][8],
                                         unsigned (&mbp)[4], unsigned (&mbn)[4], const uint32_t* st_prev, const uint32_t* st,
                                         int b2_01, int b2_23, int lane16, BSel bsel) {
  const __amdgpu_buffer_rsrc_t rp3 = panel_rsrc(st_prev, 3), rn0 = panel_rsrc(st, 0), rn1 = panel_rsrc(st, 1), rn2 = panel_rsrc(st, 2);
  constexpr int NF = 2 * R;
  constexpr int SP0 = NF - 1 < 24 ? NF - 1 : 24;   // chunk 0: the pending blocks are this chunk's k-steps 12..15 (slots >= 26)
  constexpr int OPP = epi_ops(PRELU, STASH), OPN = epi_ops(RELU, STASH);
  if constexpr (PEND)
    bf_chunk<2, R, true, SP0, OPP, STASH>(acc0, c.fr, c.rg, c.ll, c.wave, b2_01, bsel,
        [&](int k) __attribute__((always_inline)) { panel_epi<SP0, 6, PRELU, STASH>(k, acc1, in, mbp[3], rp3, lane16); });
  else
    bf_chunk<2, R, true, 0, 0, false>(acc0, c.fr, c.rg, c.ll, c.wave, b2_01, bsel, [&](int) __attribute__((always_inline)) {});
  bf_chunk<2, R, true, NF - 1, OPN, STASH>(acc1, c.fr, c.rg, c.ll, c.wave, b2_01, bsel,
      [&](int k) __attribute__((always_inline)) { panel_epi<NF - 1, 0, RELU, STASH>(k, acc0, out, mbn[0], rn0, lane16); });
  bf_chunk<2, R, true, NF - 1, OPN, STASH>(acc0, c.fr, c.rg, c.ll, c.wave, b2_23, bsel,
      [&](int k) __attribute__((always_inline)) { panel_epi<NF - 1, 2, RELU, STASH>(k, acc1, out, mbn[1], rn1, lane16); });
  bf_chunk<2, R, true, NF - 1, OPN, STASH>(acc1, c.fr, c.rg, c.ll, c.wave, b2_23, bsel,
      [&](int k) __attribute__((always_inline)) { panel_epi<NF - 1, 4, RELU, STASH>(k, acc0, out, mbn[2], rn2, lane16); });
}

// sigma activation (models.py:276-277): softplus or relu.  log1p(e), e = exp(-|x|) in (0, 1], as log(u) e / (u - 1) with u = 1 + e:
// the rounding of u cancels in the quotient (relative error ~1e-7 down to e ~ 1e-38), where log(1 + e) alone loses e below
// 6e-8 and 20 % of it at x = -15 -- and it is 6 VALU instead of libm's log1pf sequence (which cost the inference kernel its
// only scratch slot)
__device__ __forceinline__ float bf_sigma(float x, int kind) {
  if (kind != 1) return fmaxf(x, 0.f);
  const float e = __expf(-fabsf(x)), u = 1.f + e, d = u - 1.f;
  return fmaxf(x, 0.f) + (d == 0.f ? e : __logf(u) * __fdividef(e, d));
}

}  // namespace

// One workgroup (8 waves) per CU, 256 samples per workgroup iteration, one 32-sample group per wave.
template <bool STASH, bool ABN>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void nerf_mlp_fwd_bf16_kernel(const ChainFwdArgs A) {
  extern __shared__ __attribute__((aligned(16))) char bf_lds[];
  const int lane0 = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = (int)(blockDim.x >> 6);   // 8; 4 when eight-wave iterations would leave more than half of the CUs idle (launch_chain_fwd_bf16)
  const int niter = (A.rows + 255) / 256 * (8 / nw);   // the stash holds whole multiples of 8 groups, all of them read by wgrad_bf16: every one is written
  const bf16x8 bias_op = as_bf16x8(0x3F803F80u, 0u, 0u, 0u);   // B = 1 in k-slots 0, 1 (bias hi + lo)

  ChainCtx c;
  chain_start(c, bf_lds, A.wpk, FW_TOTAL, FW_L0, FW_L0, lane0, wave, nw);

#pragma unroll 1
  for (int it = blockIdx.x; it < niter; it += gridDim.x) {
    int lo = lane0;
    asm volatile("" : "+v"(lo));   // per-iteration opaque lane: nothing derived from it is hoisted out of the loop (and spilled)
    const int lane = lo, n = lane & 31, h = lane >> 5;
    const int lane16 = lane * 16;
    const int row = (it * nw + wave) * 32 + n;
    const int rc = row < A.rows ? row : A.rows - 1;
    float x[3];
    if (A.points) {
      x[0] = A.points[3 * rc]; x[1] = A.points[3 * rc + 1]; x[2] = A.points[3 * rc + 2];
    } else {
      const int ray = rc / A.S;
      const float z = A.zvals[rc];
#pragma unroll
      for (int k = 0; k < 3; ++k) x[k] = __fadd_rn(A.origins[3 * ray + k], __fmul_rn(z, A.directions[3 * ray + k]));
    }
    const size_t gidx = (size_t)it * nw + wave;   // this wave's group
    // SinusoidalEncoder (modules.py:213-228) in fp32, packed straight into B-operand registers
    unsigned pe[2][8];
    {
      const float half_pi = 1.57079632679489661923f;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          float v[2];
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int r16 = 2 * q + t;                                  // accumulator-style register index 4j + i
            const int e = 32 * b + 8 * (r16 >> 2) + 4 * h + (r16 & 3);   // posenc feature
            float val = 0.f;
            if (e < 3) {
              val = e == 0 ? x[0] : e == 1 ? x[1] : x[2];
            } else if (e < A.P) {
              const int idx = e - 3, f = idx / 6, rem = idx - 6 * f, cc = rem >= 3 ? rem - 3 : rem;
              const float a = __fmul_rn(cc == 0 ? x[0] : cc == 1 ? x[1] : x[2], (float)(1 << f));
              val = __sinf(rem >= 3 ? __fadd_rn(a, half_pi) : a);   // v_sin_f32: ~1e-6 abs, far below the bf16 rounding that follows
            }
            v[t] = val;
          }
          pe[b][q] = pack_bf16(v[0], v[1]);
        }
      if constexpr (STASH) {
        const __amdgpu_buffer_rsrc_t rp = panel_rsrc(A.bst.pe + gidx * 2 * BF_BLOCK_DW, 0);
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int jp = 0; jp < 2; ++jp)
            bf_store16(rp, lane16 + (b * 2 + jp) * BF_KB, pe[b][4 * jp], pe[b][4 * jp + 1], pe[b][4 * jp + 2], pe[b][4 * jp + 3]);
      }
    }

    unsigned ua[8][8], ub[8][8];
    unsigned m0[4] = {0u, 0u, 0u, 0u}, m1[4] = {0u, 0u, 0u, 0u};
    f32x16 acc0[2], acc1[2];
    auto hst = [&](int l) __attribute__((always_inline)) { return A.bst.h + ((size_t)l * A.bst.ngroups + gidx) * 8 * BF_BLOCK_DW; };
    auto store_bits = [&](int l, unsigned (&mb)[4]) __attribute__((always_inline)) {
      if constexpr (STASH) {
        const u32x4v q = {mb[0], mb[1], mb[2], mb[3]};
        __builtin_nontemporal_store(q, reinterpret_cast<u32x4v*>(A.bst.bits) + ((size_t)l * A.bst.ngroups + gidx) * 64 + lane);
        mb[0] = mb[1] = mb[2] = mb[3] = 0u;
      }
    };
    // ---- trunk: the packed activations alternate between ua and ub ----
    layer256<5, false, true, true, STASH>(c, acc0, acc1, ua, ua, m1, m0, hst(0), hst(0), FW_L0, FW_T, lane16,
        [&](int r) __attribute__((always_inline)) { return r == 0 ? bias_op : BF_ROWS(pe, 1); });                           // L0: pe -> ua
    // Layers 1..3 and 5..7 are the same code (ua -> ub -> ua -> ub) on other stash slots.  TRAINING: one copy, run twice, the skip
    // layer between the two rounds.  The unrolled kernel is ~100 KiB of instructions and a training launch runs 1-6 iterations
    // per workgroup: every launch streams its code through a cold instruction cache (SQC_ICACHE_MISSES_DUPLICATE = 12 % of the
    // fetches, profiles/r04_train_bf16_pmc_icache.md; the 1-iteration coarse launch takes 80-120 us against 53 us per iteration in
    // steady state).  A third less code: coarse forward -13 %, dgrad -3 %.  INFERENCE (16+ iterations per workgroup, 0.9 % duplicate
    // misses) keeps the unrolled form: rolled, hipcc spills 46 VGPRs there and the forward is 8 % slower (round-4 experiment, git history).
    if constexpr (STASH) {
#pragma unroll 1
    for (int t = 0; t < 2; ++t) {
      const int l0 = 4 * t;
      layer256<17, true, true, true, STASH>(c, acc0, acc1, ua, ub, m0, m1, hst(l0), hst(l0 + 1), FW_T, FW_T, lane16,
          [&](int r) __attribute__((always_inline)) { return r == 0 ? bias_op : BF_ROWS(ua, 1); });
      store_bits(l0, m0);
      layer256<17, true, true, true, STASH>(c, acc0, acc1, ub, ua, m1, m0, hst(l0 + 1), hst(l0 + 2), FW_T, FW_T, lane16,
          [&](int r) __attribute__((always_inline)) { return r == 0 ? bias_op : BF_ROWS(ub, 1); });
      store_bits(l0 + 1, m1);
      layer256<17, true, true, true, STASH>(c, acc0, acc1, ua, ub, m0, m1, hst(l0 + 2), hst(l0 + 3), FW_T, t == 0 ? FW_S : FW_T, lane16,
          [&](int r) __attribute__((always_inline)) { return r == 0 ? bias_op : BF_ROWS(ua, 1); });
      store_bits(l0 + 2, m0);
      if (t == 0) {
        layer256<21, true, true, true, STASH>(c, acc0, acc1, ub, ua, m1, m0, hst(3), hst(4), FW_S, FW_T, lane16,                    // skip: [h, posenc]
            [&](int r) __attribute__((always_inline)) { return r == 0 ? bias_op : r <= 16 ? BF_ROWS(ub, 1) : BF_ROWS(pe, 17); });
        store_bits(3, m1);
      }
    }
    } else {
    layer256<17, true, true, true, STASH>(c, acc0, acc1, ua, ub, m0, m1, hst(0), hst(1), FW_T, FW_T, lane16,
        [&](int r) __attribute__((always_inline)) { return r == 0 ? bias_op : BF_ROWS(ua, 1); });
    store_bits(0, m0);
    layer256<17, true, true, true, STASH>(c, acc0, acc1, ub, ua, m1, m0, hst(1), hst(2), FW_T, FW_T, lane16,
        [&](int r) __attribute__((always_inline)) { return r == 0 ? bias_op : BF_ROWS(ub, 1); });
    store_bits(1, m1);
    layer256<17, true, true, true, STASH>(c, acc0, acc1, ua, ub, m0, m1, hst(2), hst(3), FW_T, FW_S, lane16,
        [&](int r) __attribute__((always_inline)) { return r == 0 ? bias_op : BF_ROWS(ua, 1); });
    store_bits(2, m0);
    layer256<21, true, true, true, STASH>(c, acc0, acc1, ub, ua, m1, m0, hst(3), hst(4), FW_S, FW_T, lane16,                      // skip: [h, posenc]
        [&](int r) __attribute__((always_inline)) { return r == 0 ? bias_op : r <= 16 ? BF_ROWS(ub, 1) : BF_ROWS(pe, 17); });
    store_bits(3, m1);
    layer256<17, true, true, true, STASH>(c, acc0, acc1, ua, ub, m0, m1, hst(4), hst(5), FW_T, FW_T, lane16,
        [&](int r) __attribute__((always_inline)) { return r == 0 ? bias_op : BF_ROWS(ua, 1); });
    store_bits(4, m0);
    layer256<17, true, true, true, STASH>(c, acc0, acc1, ub, ua, m1, m0, hst(5), hst(6), FW_T, FW_T, lane16,
        [&](int r) __attribute__((always_inline)) { return r == 0 ? bias_op : BF_ROWS(ub, 1); });
    store_bits(5, m1);
    layer256<17, true, true, true, STASH>(c, acc0, acc1, ua, ub, m0, m1, hst(6), hst(7), FW_T, FW_T, lane16,
        [&](int r) __attribute__((always_inline)) { return r == 0 ? bias_op : BF_ROWS(ua, 1); });
    store_bits(6, m0);
    }
    // ---- bottleneck (linear): h8 = ub -> ua; its chunks 2, 3 prefetch the alpha chunk and the rgb branch ----
    const uint32_t* bnb = A.bst.bn + gidx * 8 * BF_BLOCK_DW;
    unsigned mdummy = 0u;
    {
      constexpr int NF = 34;
      const __amdgpu_buffer_rsrc_t r73 = panel_rsrc(hst(7), 3), rb0 = panel_rsrc(bnb, 0), rb1 = panel_rsrc(bnb, 1), rb2 = panel_rsrc(bnb, 2);
      auto bs = [&](int r) __attribute__((always_inline)) { return r == 0 ? bias_op : BF_ROWS(ub, 1); };
      bf_chunk<2, 17, true, 24, epi_ops(true, STASH), STASH>(acc0, c.fr, c.rg, c.ll, wave, FW_T, bs,
          [&](int k) __attribute__((always_inline)) { panel_epi<24, 6, true, STASH>(k, acc1, ub, m1[3], r73, lane16); });
      store_bits(7, m1);
      bf_chunk<2, 17, true, NF - 1, 1, STASH>(acc1, c.fr, c.rg, c.ll, wave, FW_T, bs,
          [&](int k) __attribute__((always_inline)) { panel_epi<NF - 1, 0, false, STASH>(k, acc0, ua, mdummy, rb0, lane16); });
      bf_chunk<2, 17, true, NF - 1, 1, STASH>(acc0, c.fr, c.rg, c.ll, wave, FW_AL, bs,
          [&](int k) __attribute__((always_inline)) { panel_epi<NF - 1, 2, false, STASH>(k, acc1, ua, mdummy, rb1, lane16); });
      bf_chunk<2, 17, true, NF - 1, 1, STASH>(acc1, c.fr, c.rg, c.ll, wave, FW_RG, bs,
          [&](int k) __attribute__((always_inline)) { panel_epi<NF - 1, 4, false, STASH>(k, acc0, ua, mdummy, rb2, lane16); });
    }
    // ---- alpha head: one block on h8 (row 0 = its bias; feature 0 = the raw density).  use_alpha_condition (modules.py:152-157):
    //      on the BOTTLENECK instead (+ the per-ray appearance-code term, float32); its blocks 6, 7 are still pending, their 16
    //      units ride in slots 1..12 and k-steps 13..16 read them.  ABN is a template parameter: the default kernels are at 256 VGPRs with
    //      nothing to spare (a run-time choice here cost them 1-6 spilled registers) ----
    float alpha_raw;
    {
      f32x16 aa[1];
      const __amdgpu_buffer_rsrc_t rb3 = panel_rsrc(bnb, 3);
      constexpr int ASP = ABN ? 12 : 16;
      bf_chunk<1, 17, true, ASP, 1, STASH>(aa, c.fr, c.rg, c.ll, wave, FW_RG,
          [&](int r) __attribute__((always_inline)) { return r == 0 ? bias_op : ABN ? BF_ROWS(ua, 1) : BF_ROWS(ub, 1); },
          [&](int k) __attribute__((always_inline)) { panel_epi<ASP, 6, false, STASH>(k, acc1, ua, mdummy, rb3, lane16); });
      alpha_raw = aa[0][0];
      if constexpr (ABN) alpha_raw += A.alpha_ct[min(rc / A.S, A.B - 1)];
    }
    // ---- rgb branch: hidden 256 -> 128 (+ the fp32 per-ray condition term incl. bias), ReLU ----
    unsigned rh[4][8];
    const float* ct = A.condterm + (size_t)min(rc / A.S, A.B - 1) * RGB_W + 4 * h;
    const uint32_t* rgb_ = A.bst.rgbh + gidx * 4 * BF_BLOCK_DW;
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
      const __amdgpu_buffer_rsrc_t rg0 = panel_rsrc(rgb_, 0);
      auto bs = [&](int r) __attribute__((always_inline)) { return BF_ROWS(ua, 0); };
      bf_chunk<2, 16, true, 0, 0, false>(acc0, c.fr, c.rg, c.ll, wave, FW_LG, bs, [&](int) __attribute__((always_inline)) {});
      add_ct(acc0, 0);
      bf_chunk<2, 16, true, 31, epi_ops(true, STASH), STASH>(acc1, c.fr, c.rg, c.ll, wave, FW_L0, bs,
          [&](int k) __attribute__((always_inline)) { panel_epi<31, 0, true, STASH>(k, acc0, rh, m0[0], rg0, lane16); });
      add_ct(acc1, 2);
    }
    // ---- rgb logits: 128 -> 3 (one block), sigmoid; the chain restarts (layer 0 of the next iteration is being copied);
    //      the pending panel (rgb hidden blocks 2, 3) is this chunk's k-steps 4..7: its 16 units ride in slots 1..4 ----
    {
      f32x16 lg[1];
      const __amdgpu_buffer_rsrc_t rg1 = panel_rsrc(rgb_, 1);
      bf_chunk<1, 9, true, 4, epi_ops(true, STASH), STASH>(lg, c.fr, c.rg, c.ll, wave, FW_L0,
          [&](int r) __attribute__((always_inline)) { return r == 0 ? bias_op : BF_ROWS(rh, 1); },
          [&](int k) __attribute__((always_inline)) { panel_epi<4, 2, true, STASH>(k, acc1, rh, m0[1], rg1, lane16); });
      if constexpr (STASH) {
        const u32x4v q = {m0[0], m0[1], 0u, 0u};
        __builtin_nontemporal_store(q, reinterpret_cast<u32x4v*>(A.bst.bits) + ((size_t)8 * A.bst.ngroups + gidx) * 64 + lane);
      }
      if (h == 0 && row < A.rows) {
        float4 o;
        o.x = 1.f / (1.f + expf(-lg[0][0]));
        o.y = 1.f / (1.f + expf(-lg[0][1]));
        o.z = 1.f / (1.f + expf(-lg[0][2]));
        float araw = alpha_raw;
        if (A.noise_std > 0.f)   // model_utils.noise_regularize (model_utils.py:266-282)
          araw += A.noise_std * (A.noise ? A.noise[row]
                                         : philox_normal(A.dyn ? A.dyn->rng_seed : A.noise_seed, A.dyn ? A.dyn->rng_offset : A.noise_offset, A.noise_stream, (uint32_t)row));
        o.w = bf_sigma(araw, A.sigma_act);
        A.out4[row] = o;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// data-gradient chain (training): d raw (rgb, sigma) -> dpre of every layer, stored for wgrad_bf16.hip
// ---------------------------------------------------------------------------------------------
namespace {
// dgrad weight stream, chunks in execution order (panel = 2 output blocks unless noted):
//   G1  logit^T   K = 3 of one k-step (+ a zero k-step), all 4 blocks of the rgb hidden layer in one panel      8 KiB
//   G2  rgbh^T    K = 128 (8 k-steps) -> 256, row 0 = the alpha row under use_alpha_condition (else zeros)       4 x 18
//   G3  bn^T      K = 256 -> 256, row 0 = the alpha row (B = d sigma; zeros under use_alpha_condition)            4 x 34
//   L7..L1        K = 256 -> 256                                                                                  7 x 4 x 32
//   warp on:  P0  W0^T  256 -> 64 (d posenc through layer 0),  P4  W4[256:]^T  256 -> 64 (skip rows, accumulates)  2 x 32
constexpr int DG1 = 8 * BF_KB, DG2 = 18 * BF_KB, DG3 = 34 * BF_KB, DGL = 32 * BF_KB, DGP = 32 * BF_KB;
constexpr int BW_TOTAL = DG1 + 4 * DG2 + 4 * DG3 + 28 * DGL;
static_assert(BW_TOTAL == BF_BWD_STREAM_KB * BF_KB && BW_TOTAL + 2 * DGP == BF_BWD_STREAM_DPTS_KB * BF_KB, "dgrad stream length (nrf_internal.h)");

// Backward units: accumulators -> bf16 pairs -> (MASK: zero where the stashed ReLU bit is clear) -> out, 1 KiB store per half block
template <int SPAN, int O0, bool MASK, int PO = 0, int NBLK, int NP>
__device__ __forceinline__ void panel_epi_bwd(int k, const f32x16 (&pend)[NP], unsigned (&out)[NBLK][8], unsigned mb,
                                              __amdgpu_buffer_rsrc_t rs, int lane16) {
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    if (epi_slot(u, 16, SPAN) != k) continue;
    const int o = u >> 3, q = u & 7;
    unsigned pk = pack_bf16(pend[PO + o][2 * q], pend[PO + o][2 * q + 1]);
    if (MASK) pk = bits_mask(pk, mb, u);
    out[O0 + o][q] = pk;
    if ((q & 3) == 3) {
      const int jp = q >> 2;
      bf_store16(rs, lane16 + (o * 2 + jp) * BF_KB, out[O0 + o][4 * jp], out[O0 + o][4 * jp + 1], out[O0 + o][4 * jp + 2],
                 out[O0 + o][4 * jp + 3]);
    }
  }
}
constexpr int BW_OPS_MASK = 4, BW_OPS_LIN = 1;   // VALU per unit: pack (+ shift, shift, and)

// One 256 -> 256 layer of the dgrad chain: in = dpre_l (blocks 6, 7 from acc1 during chunk 0), out = masked d h_{l-1} = dpre_{l-1}
// blocks 0..5, blocks 6, 7 pending in acc1.  PMASK / mbp3 / st_prev: the previous GEMM's epilogue (its last panel); mb / st: this one's.
template <int R, bool PMASK, class BSel>
__device__ __forceinline__ void layer256_bwd(ChainCtx& c, f32x16 (&acc0)[2], f32x16 (&acc1)[2], unsigned (&in)[8][8], unsigned (&out)[8][8],
                                             unsigned mbp3, const u32x4v& mb, const uint32_t* st_prev, const uint32_t* st,
                                             int b2_01, int b2_2, int b2_3, int lane16, BSel bsel) {
  const __amdgpu_buffer_rsrc_t rp3 = panel_rsrc(st_prev, 3), rn0 = panel_rsrc(st, 0), rn1 = panel_rsrc(st, 1), rn2 = panel_rsrc(st, 2);
  constexpr int NF = 2 * R;
  constexpr int SP0 = 22;   // chunk 0: the pending blocks are this chunk's k-steps 12..15 (slots >= 24 without a bias row)
  bf_chunk<2, R, true, SP0, PMASK ? BW_OPS_MASK : BW_OPS_LIN, true>(acc0, c.fr, c.rg, c.ll, c.wave, b2_01, bsel,
      [&](int k) __attribute__((always_inline)) { panel_epi_bwd<SP0, 6, PMASK>(k, acc1, in, mbp3, rp3, lane16); });
  bf_chunk<2, R, true, NF - 1, BW_OPS_MASK, true>(acc1, c.fr, c.rg, c.ll, c.wave, b2_01, bsel,
      [&](int k) __attribute__((always_inline)) { panel_epi_bwd<NF - 1, 0, true>(k, acc0, out, mb.x, rn0, lane16); });
  bf_chunk<2, R, true, NF - 1, BW_OPS_MASK, true>(acc0, c.fr, c.rg, c.ll, c.wave, b2_2, bsel,
      [&](int k) __attribute__((always_inline)) { panel_epi_bwd<NF - 1, 2, true>(k, acc1, out, mb.y, rn1, lane16); });
  bf_chunk<2, R, true, NF - 1, BW_OPS_MASK, true>(acc1, c.fr, c.rg, c.ll, c.wave, b2_3, bsel,
      [&](int k) __attribute__((always_inline)) { panel_epi_bwd<NF - 1, 4, true>(k, acc0, out, mb.z, rn2, lane16); });
}
}  // namespace

// Both levels in ONE launch: workgroups [0, n0) walk level 0, the rest level 1, split in proportion to the levels'
// iteration counts, so every workgroup runs the same number of iterations of ONE level (its weight stream never switches).
// The level's arguments are indexed in the kernarg segment (scalar loads).
struct ChainBwdBf16Args2 { ChainBwdBf16Args a[2]; int n0; };
template <bool DPTS>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void nerf_mlp_bwd_bf16_kernel(const ChainBwdBf16Args2 P) {
  const int lvl = (int)blockIdx.x >= P.n0 ? 1 : 0;
  const ChainBwdBf16Args& A = P.a[lvl];
  const int wg0 = lvl ? (int)blockIdx.x - P.n0 : (int)blockIdx.x;        // this workgroup's index inside its level
  const int wgn = lvl ? (int)gridDim.x - P.n0 : P.n0;                    // workgroups of its level
  extern __shared__ __attribute__((aligned(16))) char bf_lds[];
  const int lane0 = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = (int)(blockDim.x >> 6);   // as the forward kernel
  const int niter = (A.rows + 255) / 256 * (8 / nw);   // the stash holds whole multiples of 8 groups, all of them read by wgrad_bf16: every one is written
  const BfStash& S = A.st;

  ChainCtx c;
  chain_start(c, bf_lds, A.wpk, BW_TOTAL + (DPTS ? 2 * DGP : 0), DG1, DG2, lane0, wave, nw);

#pragma unroll 1
  for (int it = wg0; it < niter; it += wgn) {
    int lo = lane0;
    asm volatile("" : "+v"(lo));   // per-iteration opaque lane (see the forward kernel)
    const int lane = lo, n = lane & 31, h = lane >> 5;
    const int lane16 = lane * 16;
    const size_t gidx = (size_t)it * nw + wave;
    const int row = (it * nw + wave) * 32 + n;
    float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < A.rows) d = A.d_raw4[row];
    auto bits_of = [&](int l) __attribute__((always_inline)) {   // nrf_internal.h BfStash::bits
      return __builtin_nontemporal_load(reinterpret_cast<const u32x4v*>(S.bits) + ((size_t)l * S.ngroups + gidx) * 64 + lane);
    };
    auto dyst = [&](int l) __attribute__((always_inline)) { return S.dy + ((size_t)l * S.ngroups + gidx) * 8 * BF_BLOCK_DW; };

    // ---- d raw -> the "small" dY block (features 0..3 = d rgb logits, d raw sigma) ----
    unsigned dsm[4];
    dsm[0] = h == 0 ? pack_bf16(d.x, d.y) : 0u;
    dsm[1] = h == 0 ? pack_bf16(d.z, d.w) : 0u;
    dsm[2] = dsm[3] = 0u;
    {
      const __amdgpu_buffer_rsrc_t rsm = panel_rsrc(S.dsmall + gidx * 2 * BF_BLOCK_DW, 0);
      bf_store16(rsm, lane16, dsm[0], dsm[1], 0u, 0u);
#pragma unroll
      for (int i = 1; i < 4; ++i) bf_store16(rsm, lane16 + i * BF_KB, 0u, 0u, 0u, 0u);
    }
    const unsigned dsig2 = pack_bf16(d.w, d.w);
    const u32x4v mq8 = bits_of(8);

    unsigned ua[8][8], ub[8][8];
    f32x16 acc0[2], acc1[2];
    // ---- G1: d rgb hidden = W_logit . d logits (K index 3 = d sigma meets a zero weight row), ReLU mask of the rgb hidden layer;
    //      one panel of 4 blocks; its epilogue runs behind it (G2's first k-step already needs block 0) ----
    unsigned drg[4][8];
    {
      f32x16 g1[4];
      bf_chunk<4, 2, true, 0, 0, false>(g1, c.fr, c.rg, c.ll, wave, DG2,
          [&](int r) __attribute__((always_inline)) { return as_bf16x8(dsm[0], dsm[1], dsm[2], dsm[3]); },
          [&](int) __attribute__((always_inline)) {});
      const uint32_t* gb = S.drgbh + gidx * 4 * BF_BLOCK_DW;
      const __amdgpu_buffer_rsrc_t r0 = panel_rsrc(gb, 0), r1 = panel_rsrc(gb, 1);
#pragma unroll
      for (int k = 1; k <= 16; ++k) panel_epi_bwd<16, 0, true, 0>(k, g1, drg, mq8.x, r0, lane16);
#pragma unroll
      for (int k = 1; k <= 16; ++k) panel_epi_bwd<16, 2, true, 2>(k, g1, drg, mq8.y, r1, lane16);
    }
    // ---- G2: d bottleneck = W_rgbh[0:256] . d rgb hidden (+ w_alpha d sigma, row 0, when the alpha head reads the bottleneck):
    //      linear -> ub blocks 0..5, blocks 6, 7 pending ----
    const uint32_t* dbn = S.dbn + gidx * 8 * BF_BLOCK_DW;
    {
      const __amdgpu_buffer_rsrc_t rn0 = panel_rsrc(dbn, 0), rn1 = panel_rsrc(dbn, 1), rn2 = panel_rsrc(dbn, 2);
      auto bs = [&](int r) __attribute__((always_inline)) { return r == 0 ? as_bf16x8(dsig2, 0u, 0u, 0u) : BF_ROWS(drg, 1); };
      bf_chunk<2, 9, true, 0, 0, false>(acc0, c.fr, c.rg, c.ll, wave, DG2, bs, [&](int) __attribute__((always_inline)) {});
      bf_chunk<2, 9, true, 17, BW_OPS_LIN, true>(acc1, c.fr, c.rg, c.ll, wave, DG2, bs,
          [&](int k) __attribute__((always_inline)) { panel_epi_bwd<17, 0, false>(k, acc0, ub, 0u, rn0, lane16); });
      bf_chunk<2, 9, true, 17, BW_OPS_LIN, true>(acc0, c.fr, c.rg, c.ll, wave, DG3, bs,
          [&](int k) __attribute__((always_inline)) { panel_epi_bwd<17, 2, false>(k, acc1, ub, 0u, rn1, lane16); });
      bf_chunk<2, 9, true, 17, BW_OPS_LIN, true>(acc1, c.fr, c.rg, c.ll, wave, DG3, bs,
          [&](int k) __attribute__((always_inline)) { panel_epi_bwd<17, 4, false>(k, acc0, ub, 0u, rn2, lane16); });
    }
    // ---- G3: d h8 = W_bn . d bottleneck + w_alpha d sigma (row 0), mask of layer 7 -> dpre_7: ub -> ua ----
    u32x4v mq = bits_of(7);
    {
      constexpr int NF = 34;
      const __amdgpu_buffer_rsrc_t rp3 = panel_rsrc(dbn, 3);
      const uint32_t* st = dyst(7);
      const __amdgpu_buffer_rsrc_t rn0 = panel_rsrc(st, 0), rn1 = panel_rsrc(st, 1), rn2 = panel_rsrc(st, 2);
      auto bs = [&](int r) __attribute__((always_inline)) { return r == 0 ? as_bf16x8(dsig2, 0u, 0u, 0u) : BF_ROWS(ub, 1); };
      bf_chunk<2, 17, true, 24, BW_OPS_LIN, true>(acc0, c.fr, c.rg, c.ll, wave, DG3, bs,
          [&](int k) __attribute__((always_inline)) { panel_epi_bwd<24, 6, false>(k, acc1, ub, 0u, rp3, lane16); });
      bf_chunk<2, 17, true, NF - 1, BW_OPS_MASK, true>(acc1, c.fr, c.rg, c.ll, wave, DG3, bs,
          [&](int k) __attribute__((always_inline)) { panel_epi_bwd<NF - 1, 0, true>(k, acc0, ua, mq.x, rn0, lane16); });
      bf_chunk<2, 17, true, NF - 1, BW_OPS_MASK, true>(acc0, c.fr, c.rg, c.ll, wave, DGL, bs,
          [&](int k) __attribute__((always_inline)) { panel_epi_bwd<NF - 1, 2, true>(k, acc1, ua, mq.y, rn1, lane16); });
      bf_chunk<2, 17, true, NF - 1, BW_OPS_MASK, true>(acc1, c.fr, c.rg, c.ll, wave, DGL, bs,
          [&](int k) __attribute__((always_inline)) { panel_epi_bwd<NF - 1, 4, true>(k, acc0, ua, mq.z, rn2, lane16); });
    }
    // ---- l = 7..1: d h_l = W_l[0:256] . dpre_l, mask of layer l-1 -> dpre_{l-1}; the arrays alternate ----
    // after L1 the stream continues with the d posenc GEMMs (DPTS) or restarts with G1, G2
    constexpr int AFTER_A = DPTS ? DGP : DG1, AFTER_B = DPTS ? DGP : DG2;
#define BW_LAYER(L, IN, OUT, B2, B3)                                                                                        \
    {                                                                                                                        \
      const unsigned mbp3 = mq.w;                                                                                            \
      mq = bits_of((L) - 1);                                                                                                 \
      layer256_bwd<16, true>(c, acc0, acc1, IN, OUT, mbp3, mq, dyst(L), dyst((L) - 1), DGL, B2, B3, lane16,                  \
          [&](int r) __attribute__((always_inline)) { return BF_ROWS(IN, 0); });                                            \
    }
#pragma unroll 1
    for (int t = 0; t < 3; ++t) {   // (7, 6), (5, 4), (3, 2): ONE copy of the layer pair (code size: see the forward kernel)
      BW_LAYER(7 - 2 * t, ua, ub, DGL, DGL)
      BW_LAYER(6 - 2 * t, ub, ua, DGL, DGL)
    }
    BW_LAYER(1, ua, ub, AFTER_A, AFTER_B)
#undef BW_LAYER
    // dpre_0 = ub: blocks 0..5 stored, blocks 6, 7 pending in acc1 (mask word mq.w, stash dyst(0) panel 3)
    if constexpr (!DPTS) {
      const __amdgpu_buffer_rsrc_t rp3 = panel_rsrc(dyst(0), 3);
#pragma unroll
      for (int k = 1; k <= 16; ++k) panel_epi_bwd<16, 6, true>(k, acc1, ub, mq.w, rp3, lane16);
    } else {
      // ---- warp on: d posenc = W0 . dpre_0 + W4[256:] . dpre_4 (two 256 -> 64 GEMMs into one accumulator panel), chain rule
      //      through SinusoidalEncoder (modules.py:213-228) -> d points, float32, for the warp field's backward ----
      const __amdgpu_buffer_rsrc_t rp3 = panel_rsrc(dyst(0), 3);
      // dpre_4 back from its stash (stored four layers ago by this wave; the waits since have retired it).  ua is free: the first
      // half of the loads flies under P0, the second under the first half of P4 (all 64 registers at once next to ub, both
      // accumulator panels and the fragment ring is what made this variant spill)
      const u32x4v* d4src = reinterpret_cast<const u32x4v*>(dyst(SKIP_LAYER)) + lane;
      auto load_d4 = [&](int b0) __attribute__((always_inline)) {
#pragma unroll
        for (int b = b0; b < b0 + 4; ++b)
#pragma unroll
          for (int jp = 0; jp < 2; ++jp) {
            const u32x4v q = d4src[(b * 2 + jp) * 64];
            ua[b][4 * jp] = q.x; ua[b][4 * jp + 1] = q.y; ua[b][4 * jp + 2] = q.z; ua[b][4 * jp + 3] = q.w;
          }
      };
      load_d4(0);
      bf_chunk<2, 16, true, 22, BW_OPS_MASK, true>(acc0, c.fr, c.rg, c.ll, wave, DG1,
          [&](int r) __attribute__((always_inline)) { return BF_ROWS(ub, 0); },
          [&](int k) __attribute__((always_inline)) { panel_epi_bwd<22, 6, true>(k, acc1, ub, mq.w, rp3, lane16); });
      load_d4(4);
      bf_chunk<2, 16, false, 0, 0, false>(acc0, c.fr, c.rg, c.ll, wave, DG2,
          [&](int r) __attribute__((always_inline)) { return BF_ROWS(ua, 0); }, [&](int) __attribute__((always_inline)) {});
      const int r = row < A.rows ? row : A.rows - 1;
      const float x[3] = {A.points[3 * r], A.points[3 * r + 1], A.points[3 * r + 2]};
      const float half_pi = 1.57079632679489661923f;
      float dx[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
          const int e = 32 * o + 8 * (rr >> 2) + 4 * h + (rr & 3);   // posenc feature of this accumulator register
          const float g = acc0[o][rr];
          if (e < 3) {
            dx[0] += e == 0 ? g : 0.f; dx[1] += e == 1 ? g : 0.f; dx[2] += e == 2 ? g : 0.f;
          } else if (e < A.P) {
            const int idx = e - 3, f = idx / 6, rem = idx - 6 * f, cc = rem >= 3 ? rem - 3 : rem;
            const float fr = (float)(1 << f);
            const float a = __fmul_rn(cc == 0 ? x[0] : cc == 1 ? x[1] : x[2], fr);
            // d sin(a) = fr cos(a) = fr sin(a + pi/2);  d sin(a + pi/2) = fr sin(a + pi)
            const float dv = fr * __sinf(rem >= 3 ? __fadd_rn(a, 2.f * half_pi) : __fadd_rn(a, half_pi)) * g;
            dx[0] += cc == 0 ? dv : 0.f; dx[1] += cc == 1 ? dv : 0.f; dx[2] += cc == 2 ? dv : 0.f;
          }
        }
#pragma unroll
      for (int k = 0; k < 3; ++k) dx[k] += __shfl_xor(dx[k], 32);   // the two lane halves hold different features of the sample
      if (h == 0 && row < A.rows_pad) {
        float* o = A.d_points + (size_t)row * 3;
        o[0] = dx[0]; o[1] = dx[1]; o[2] = dx[2];
      }
    }
  }
}

// Waves per workgroup of a chain launch: 8 (two per SIMD, the design point); 4 when the eight-wave iterations would give at most half
// of the CUs a workgroup -- a 256-row iteration is MFMA-bound at ~30 of its ~50 us, so halving its rows and spreading them over
// twice the CUs nearly halves such a launch (the 128-ray share of a 1024-ray batch: bench.py --rays-per-gpu 128)
static int chain_waves(int iters8, int max_grid) { return 2 * iters8 <= max_grid ? 4 : 8; }

void launch_chain_bwd_bf16(const ChainBwdBf16Args& a0, const ChainBwdBf16Args* a1, int max_grid, hipStream_t stream) {
  const size_t lds = BF_LDS_BYTES;
  const int i0 = (a0.rows + 255) / 256, i1 = a1 ? (a1->rows + 255) / 256 : 0;
  const int nw = chain_waves(i0 + i1, max_grid);
  const int it0 = i0 * (8 / nw), it1 = i1 * (8 / nw);
  int grid = it0 + it1 < max_grid ? it0 + it1 : max_grid;
  int n0 = grid;
  if (a1) {   // workgroups per level in proportion to the iterations, at least one each
    n0 = (int)(((long long)grid * it0 + (it0 + it1) / 2) / (it0 + it1));
    n0 = n0 < 1 ? 1 : n0 > grid - 1 ? grid - 1 : n0;
    if (grid < 2) { grid = 2; n0 = 1; }
  }
  ChainBwdBf16Args2 p;
  p.a[0] = a0; p.a[1] = a1 ? *a1 : a0; p.n0 = n0;
  if (a0.d_points) {
    (void)hipFuncSetAttribute((const void*)nerf_mlp_bwd_bf16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(nerf_mlp_bwd_bf16_kernel<true>, dim3(grid), dim3(64 * nw), lds, stream, p);
  } else {
    (void)hipFuncSetAttribute((const void*)nerf_mlp_bwd_bf16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(nerf_mlp_bwd_bf16_kernel<false>, dim3(grid), dim3(64 * nw), lds, stream, p);
  }
}

// dray[ray][f] = sum over the ray's samples of dpre_rgbh[sample][f]  (gradient of the per-ray rgb-condition term), read back
// from the bf16 stash: [group][b (4)][jp][lane = n + 32 h][(jj, i)], feature f = 32 b + 8 (2 jp + jj) + 4 h + i.
// One workgroup per ray; thread (granule gi = (b, jp, h), sample lane sl): 16 consecutive threads read 16 consecutive samples
// of one granule column = 256 contiguous bytes; 8 float partial sums per thread, folded over the sample lanes through LDS.
__global__ __launch_bounds__(256) void dray_bf16_kernel(const uint32_t* __restrict__ drgbh, int S, float* __restrict__ dray) {
  __shared__ float sm[16][RGB_W + 1];
  const int ray = blockIdx.x, sl = threadIdx.x & 15, gi = threadIdx.x >> 4;
  const int b = gi >> 2, jp = (gi >> 1) & 1, h = gi & 1;
  const uint4* base = reinterpret_cast<const uint4*>(drgbh);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int k = sl; k < S; k += 16) {
    const size_t row = (size_t)ray * S + k;
    const uint4 v = base[(row >> 5) * (4 * BF_BLOCK_DW / 4) + (b * 2 + jp) * 64 + (int)(row & 31) + 32 * h];
    const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc[2 * i] += __uint_as_float(u[i] << 16); acc[2 * i + 1] += __uint_as_float(u[i] & 0xFFFF0000u); }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) sm[sl][32 * b + 8 * (2 * jp + (e >> 2)) + 4 * h + (e & 3)] = acc[e];
  __syncthreads();
  if (threadIdx.x < RGB_W) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += sm[q][threadIdx.x];
    dray[(size_t)ray * RGB_W + threadIdx.x] = t;
  }
}

void launch_dray_bf16(const uint32_t* drgbh, int B, int S, float* dray, hipStream_t stream) {
  hipLaunchKernelGGL(dray_bf16_kernel, dim3(B), dim3(256), 0, stream, drgbh, S, dray);
}

namespace {
// One descriptor fills rows of ONE chunk (panel) of a weight stream: [row][block of the panel][lane] x 16 B.
//   kind 0: `ngroups` k-step rows (row = 2b + s) of output blocks oblk0 .. oblk0 + nout of the GEMM; lane (m, h) gets 8 bf16:
//           slot e <-> K index 32b + 8(2s + e/4) + 4h + e%4, column 32 (oblk0 + o) + m.
//   kind 1: the bias row: slots 0/1 of the h = 0 lanes = bf16 hi / lo parts of bias[column].
// dst_off = first row written; nout_panel = blocks per row of the chunk.
__global__ __launch_bounds__(256) void bf16_pack_kernel(const RcPackDesc* __restrict__ descs, const float* __restrict__ params,
                                                        float* __restrict__ ws) {
  const RcPackDesc d = descs[blockIdx.y];
  const int total = d.ngroups * d.nout * 64;
  uint4* dst = reinterpret_cast<uint4*>(ws + d.dst_off);
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int lane = idx & 63, o = (idx >> 6) % d.nout, rowi = (idx >> 6) / d.nout;
    const int rowk = d.x3 ? rowi >> 1 : rowi, var = d.x3 ? rowi & 1 : 0;   // x3: rows (W_hi, W_lo) per k-step
    const int m = lane & 31, h = lane >> 5, b = rowk >> 1, s = rowk & 1;
    const int col = 32 * (d.oblk0 + o) + m;
    // two leaves side by side (SE3 heads): output columns (forward, bias) / K indices (transposed) >= split come from src_off2
    const bool col2 = !d.transposed && d.split > 0 && col >= d.split;
    const long long leaf = col2 ? d.src_off2 : d.src_off;
    const int lcol = col2 ? col - d.split : col;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (d.kind == 2) {   // a zero row
    } else if (d.kind == 1) {
      if (h == 0 && col < d.ncols) {
        const float bias = params[leaf + lcol];
        const float hi = __uint_as_float(pack_bf16(bias, 0.f) << 16);
        v[0] = hi; v[1] = bias - hi;
        if (d.x3) v[2] = v[1] - __uint_as_float(pack_bf16(v[1], 0.f) << 16);   // the third term: bias to 24 bits
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = 32 * b + 8 * (2 * s + (e >> 2)) + 4 * h + (e & 3);
        if (k < d.krows && col < d.ncols) {
          if (d.transposed) {
            const bool k2 = d.split > 0 && k >= d.split;
            v[e] = params[(k2 ? d.src_off2 : d.src_off) + (int64_t)(d.row0 + col) * d.src_ld + (k2 ? k - d.split : k)];
          } else {
            v[e] = params[leaf + (int64_t)(d.row0 + k) * d.src_ld + lcol];
          }
        }
      }
    }
    if (var == 1) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] -= __uint_as_float(pack_bf16(v[e], 0.f) << 16);   // the lo part
    }
    uint4 out;
    out.x = pack_bf16(v[0], v[1]); out.y = pack_bf16(v[2], v[3]); out.z = pack_bf16(v[4], v[5]); out.w = pack_bf16(v[6], v[7]);
    dst[(size_t)(rowi * d.nout_panel + d.o0 + o) * 64 + lane] = out;
  }
}
}  // namespace

void launch_bf16_pack(const RcPackDesc* descs, int ndesc, const float* params, float* ws, hipStream_t stream) {
  if (ndesc > 0) bf16_pack_kernel<<<dim3(8, ndesc), 256, 0, stream>>>(descs, params, ws);
}

namespace {
template <bool STASH, bool ABN>
void launch_fwd_variant(const ChainFwdArgs& a, int max_grid, hipStream_t stream) {
  (void)hipFuncSetAttribute((const void*)nerf_mlp_fwd_bf16_kernel<STASH, ABN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BF_LDS_BYTES);
  const int i8 = (a.rows + 255) / 256;
  const int nw = chain_waves(i8, max_grid);
  const int nit = i8 * (8 / nw);
  hipLaunchKernelGGL((nerf_mlp_fwd_bf16_kernel<STASH, ABN>), dim3(nit < max_grid ? nit : max_grid), dim3(64 * nw), BF_LDS_BYTES, stream, a);
}
}  // namespace

void launch_chain_fwd_bf16(const ChainFwdArgs& a, int max_grid, hipStream_t stream) {
  const bool stash = a.bst.h != nullptr;      // training: stash every layer's packed output + sign bits
  const bool abn = a.alpha_ct != nullptr;     // use_alpha_condition: the alpha head reads the bottleneck
  if (stash) { if (abn) launch_fwd_variant<true, true>(a, max_grid, stream); else launch_fwd_variant<true, false>(a, max_grid, stream); }
  else { if (abn) launch_fwd_variant<false, true>(a, max_grid, stream); else launch_fwd_variant<false, false>(a, max_grid, stream); }
}

}  // namespace nrf
