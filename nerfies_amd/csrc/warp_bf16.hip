// SE(3) warp field with a bf16-operand trunk (NRF_FLAG_BF16 together with the warp field; BASELINE config D names bf16 for its
// MLPs, and SE3Field's 6 x 128 trunk is the other MLP config D contains): forward, forward-mode tangent pass, and the reverse
// of both.  Same design as the NeRF chains (bf16_chain.h): transposed GEMMs, a wave owns 32 rows and all 128 features, activations
// stay in registers across layers, weights stream through the three-slot LDS ring, a layer is two panels of two output blocks
// and the epilogue of a panel rides under the MFMAs of the next one.  Both panels of a layer (and the heads / heads^T block) share
// one ring slot, i.e. one barrier per layer: 6 per 256-row iteration forward (52 KiB slots: the skip layer fits one), 5-6 reverse.
//
// Replaces, in that mode (reference /root/reference/nerfies):
//   modules.AnnealedSinusoidalEncoder   modules.py:231-294   prologue, fp32, packed to bf16 B operands
//   glo.GloEncoder                      glo.py:22-53         per-row code gather in the prologue
//   warping.SE3Field.warp / __call__    warping.py:322-389   trunk on bf16 MFMA; heads' outputs (w, v) accumulate in fp32
//   rigid_body.exp_se3                  rigid_body.py:54-89  fp32 closed form per row (se3_math.h), as in warp_chain.hip
//   jax.jacfwd(self.warp)               warping.py:385-387   tangent pass: 3 tangents per row, ReLU' = the primal sign bits
// Kept in fp32: the point, exp_se3 and its VJP, (w, v) and their tangents as the elastic kernel reads them, the GLO table and
// its gradient.  bf16: the trunk's operands (inputs incl. the point's identity features, activations, weights, dpre) and the
// stash the weight gradients are formed from (wgrad_bf16.hip).
#include "bf16_chain.h"
#include "se3_math.h"

namespace nrf {

namespace {

typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));

// chunks (= ring slots = barriers) of the forward stream: layer 0 | layers 1..3 | skip layer | layer 5 + heads
constexpr int WF_L0 = 20 * BF_KB, WF_T = 36 * BF_KB, WF_S = 52 * BF_KB, WF_HD = 9 * BF_KB, WF_T5 = WF_T + WF_HD;
constexpr int WF_TOTAL = WF_L0 + 3 * WF_T + WF_S + WF_T5;
constexpr int WF_SLOT = WF_S;   // 3 x 52 KiB = 156 of the 160 KiB
static_assert(WF_TOTAL == BFW_FWD_STREAM_KB * BF_KB, "SE3 forward stream length (nrf_internal.h)");
// reverse stream: heads^T + layer 5 | layers 4..1 | the two code-gradient GEMMs (primal only)
constexpr int WB_GH = 8 * BF_KB, WB_L = 32 * BF_KB, WB_G5 = WB_GH + WB_L;
constexpr int WB_TAN_TOTAL = WB_G5 + 4 * WB_L, WB_TOTAL = WB_TAN_TOTAL + WB_L;
static_assert(WB_TOTAL == BFW_BWD_STREAM_KB * BF_KB && WB_TAN_TOTAL == BFW_BWD_TAN_STREAM_KB * BF_KB, "SE3 reverse stream length");

enum { EP_LIN = 0, EP_RELU = 1, EP_MASK = 2 };
__device__ __forceinline__ constexpr int ep_ops(int mode, bool stash) { return mode == EP_LIN ? 1 : mode == EP_MASK ? 4 : stash ? 4 : 2; }

// Units of a pending panel that fall on slot k (cf. mlp_bf16.hip panel_epi): accumulators -> bf16 pairs -> ReLU (+ sign bits into
// mb) or mask by the bits in mb -> out[O0 + o]; STORE: 1 KiB stash store per half block
template <int SPAN, int O0, int MODE, bool STORE, int PO = 0, int NBLK, int NP>
__device__ __forceinline__ void wpanel_epi(int k, const f32x16 (&pend)[NP], unsigned (&out)[NBLK][8], unsigned& mb,
                                           __amdgpu_buffer_rsrc_t rs, int lane16) {
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    if (epi_slot(u, 16, SPAN) != k) continue;
    const int o = u >> 3, q = u & 7;
    unsigned pk = pack_bf16(pend[PO + o][2 * q], pend[PO + o][2 * q + 1]);
    if (MODE == EP_RELU) {
      pk = relu_pk(pk);
      if (STORE) mb = bits_push(mb, pk);
    } else if (MODE == EP_MASK) {
      pk = bits_mask(pk, mb, u);
    }
    out[O0 + o][q] = pk;
    if (STORE && (q & 3) == 3) {
      const int jp = q >> 2;
      bf_store16(rs, lane16 + (o * 2 + jp) * BF_KB, out[O0 + o][4 * jp], out[O0 + o][4 * jp + 1], out[O0 + o][4 * jp + 2],
                 out[O0 + o][4 * jp + 3]);
    }
  }
}

// One 128-wide layer = 2 panels in ONE chunk (fragments F0 .. F0 + 4R - 1 of a chunk of NFC; the skip layer of the forward pass
// fills a 52 KiB slot, the heads / heads^T chunks share a slot with the neighbouring layer): one barrier per layer.  in: packed
// input (its blocks 2, 3 arrive from acc1 = the previous layer's second panel during panel 0 when PEND); out: blocks 0, 1; blocks
// 2, 3 stay pending in acc1.  mbp1: the previous layer's bit word of its panel 1 (RELU: completed here; MASK: read), mbn0: this
// layer's word of panel 0.  b2: size of the chunk two ahead (used by the panel that holds the chunk's barrier).
template <int R, bool PEND, int PMODE, int MODE, bool STORE, int F0, int NFC, int SLOT, int PRE = 0, class BSel>
__device__ __forceinline__ void layer128(ChainCtx& c, f32x16 (&acc0)[2], f32x16 (&acc1)[2], unsigned (&in)[4][8], unsigned (&out)[4][8],
                                         unsigned& mbp1, unsigned& mbn0, const uint32_t* st_prev, const uint32_t* st, int b2,
                                         int lane16, BSel bsel) {
  const __amdgpu_buffer_rsrc_t rp1 = panel_rsrc(st_prev, 1), rn0 = panel_rsrc(st, 0);
  constexpr int NF = 2 * R;
  constexpr int SP0 = 6;   // panel 0: the pending blocks 2, 3 are its k-steps 4..7 (slots >= 8, >= 10 with a bias row)
  if constexpr (PEND)
    bf_panel<2, R, true, SP0, ep_ops(PMODE, STORE), STORE, F0, NFC, SLOT, PRE>(acc0, c.fr, c.rg, c.ll, c.wave, b2, bsel,
        [&](int k) __attribute__((always_inline)) { wpanel_epi<SP0, 2, PMODE, STORE>(k, acc1, in, mbp1, rp1, lane16); });
  else
    bf_panel<2, R, true, 0, 0, false, F0, NFC, SLOT, PRE>(acc0, c.fr, c.rg, c.ll, c.wave, b2, bsel, [&](int) __attribute__((always_inline)) {});
  bf_panel<2, R, true, NF - 1, ep_ops(MODE, STORE), STORE, F0 + NF, NFC, SLOT, PRE + (PEND && STORE ? 4 : 0)>(acc1, c.fr, c.rg, c.ll, c.wave, b2, bsel,
      [&](int k) __attribute__((always_inline)) { wpanel_epi<NF - 1, 0, MODE, STORE>(k, acc0, out, mbn0, rn0, lane16); });
}

// cosine_easing_window (modules.py:274-294) of band f: 0.5 (1 + cos(pi clip(alpha - f, 0, 1) + pi))
__device__ __forceinline__ float band_window(float alpha, int f) {
  const float pi = 3.14159265358979323846f;
  const float cl = fminf(fmaxf(alpha - (float)f, 0.f), 1.f);
  return 0.5f * (1.f + cosf(__fadd_rn(__fmul_rn(pi, cl), pi)));
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// forward (primal, or the tangent pass of the warp Jacobian)
// ---------------------------------------------------------------------------------------------
// Iterations (256 rows = 8 groups) [0, n0) belong to level 0, [n0, ntot) to level 1 (the background batch behind the coarse
// samples); the tangent pass runs 3 x the primal iterations, iteration = c * nit_prim + primal iteration.
struct WarpBf16FwdArgs2 { WarpFwdArgs a[2]; int n0, ntot; };
template <bool STASH, bool TANGENT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void se3_fwd_bf16_kernel(const WarpBf16FwdArgs2 P) {
  extern __shared__ __attribute__((aligned(16))) char bf_lds[];
  const int lane0 = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bf16x8 bias_op = TANGENT ? as_bf16x8(0u, 0u, 0u, 0u) : as_bf16x8(0x3F803F80u, 0u, 0u, 0u);   // the tangent carries no bias

  // cosine_easing_window of every band, once per kernel (the levels of a launch share the step's alpha), kept in SGPRs
  float wnd[10];
  {
    const float warp_alpha = P.a[0].dyn ? P.a[0].dyn->warp_alpha : P.a[0].alpha;
#pragma unroll
    for (int f = 0; f < 10; ++f) wnd[f] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(band_window(warp_alpha, f))));
  }

  ChainCtx c;
  chain_start<WF_SLOT>(c, bf_lds, P.a[0].bwpk, WF_TOTAL, WF_L0, WF_T, lane0, wave);

#pragma unroll 1
  for (int it = blockIdx.x; it < P.ntot; it += gridDim.x) {
    const int lv = it >= P.n0 ? 1 : 0;
    const WarpFwdArgs& A = P.a[lv];
    int lo = lane0;
    asm volatile("" : "+v"(lo));   // per-iteration opaque lane (mlp_bf16.hip)
    const int lane = lo, n = lane & 31, h = lane >> 5;
    const int lane16 = lane * 16;
    const int itl = it - (lv ? P.n0 : 0);                 // iteration inside the level
    const int nit_prim = (A.rows + 255) / 256;            // tangent: A.rows = the PRIMAL rows
    const int cdir = TANGENT ? itl / nit_prim : 0;        // coordinate the tangent runs along
    const int pit = TANGENT ? itl - cdir * nit_prim : itl;
    const int row = pit * 256 + wave * 32 + n;            // (primal) row of this lane
    const int rc = row < A.rows ? row : A.rows - 1;
    const size_t gprim = (size_t)pit * 8 + wave;                            // primal group
    const size_t gidx = TANGENT ? (size_t)cdir * A.bng_prim + gprim : gprim;   // this pass's group
    const int F = A.F, cbase = 3 + 6 * F;                  // first code feature
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
    if (!TANGENT && A.points_raw && h == 0 && row < A.rows) {
      A.points_raw[3 * (size_t)row] = x[0]; A.points_raw[3 * (size_t)row + 1] = x[1]; A.points_raw[3 * (size_t)row + 2] = x[2];
    }
    const float* __restrict__ code = A.embed_table + (int64_t)id * A.G;   // glo.py:50-53
    // ---- trunk input [annealed posenc(x), code] (warping.py:326-327) or its derivative along x_cdir, packed
    //      straight into B-operand registers: lane (n, h) holds features 32 b + 8 j + 4 h + i.  The feature index of a register
    //      differs between the lane halves by 4: both candidates are evaluated with a COMPILE-TIME index (band, component and phase
    //      fold to constants; the band windows sit in SGPRs) and the lane half selects ----
    unsigned win[2][8];
    {
      const float half_pi = 1.57079632679489661923f;
      auto feature = [&](int e) __attribute__((always_inline)) -> float {   // e: a constant after unrolling
        float v = 0.f;
        if (e < 3) {
          v = TANGENT ? (e == cdir ? 1.f : 0.f) : (e == 0 ? x[0] : e == 1 ? x[1] : x[2]);
        } else if (e < cbase) {
          const int idx = e - 3, f = idx / 6, rem = idx - 6 * f, cc = rem >= 3 ? rem - 3 : rem;
          const float fr = (float)(1 << f);
          const float a = __fmul_rn(cc == 0 ? x[0] : cc == 1 ? x[1] : x[2], fr);
          const float wdw = wnd[f < 10 ? f : 9];
          if (TANGENT)   // d(w sin a) = w f cos a = w f sin(a + pi/2);  d(w sin(a + pi/2)) = w f sin(a + pi)
            v = cc == cdir ? wdw * fr * __sinf(__fadd_rn(a, rem >= 3 ? 2.f * half_pi : half_pi)) : 0.f;
          else
            v = wdw * __sinf(rem >= 3 ? __fadd_rn(a, half_pi) : a);
        } else if (e < cbase + A.G) {
          v = TANGENT ? 0.f : code[e - cbase];
        }
        return v;
      };
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int r16 = 2 * q, e0 = 32 * b + 8 * (r16 >> 2) + (r16 & 3);
          const float lo0 = feature(e0), lo1 = feature(e0 + 1), hi0 = feature(e0 + 4), hi1 = feature(e0 + 5);
          win[b][q] = pack_bf16(h ? hi0 : lo0, h ? hi1 : lo1);
        }
      if constexpr (STASH) {
        const __amdgpu_buffer_rsrc_t rp = panel_rsrc(A.bst.win + gidx * 2 * BF_BLOCK_DW, 0);
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int jp = 0; jp < 2; ++jp)
            bf_store16(rp, lane16 + (b * 2 + jp) * BF_KB, win[b][4 * jp], win[b][4 * jp + 1], win[b][4 * jp + 2], win[b][4 * jp + 3]);
      }
    }

    unsigned ua[4][8], ub[4][8];
    f32x16 acc0[2], acc1[2];
    auto hst = [&](int l) __attribute__((always_inline)) { return A.bst.h + ((size_t)l * A.bst.ngroups + gidx) * 4 * BF_BLOCK_DW; };
    // ReLU bits: the primal pass builds them (2 dwords per lane and layer), the tangent pass reads the primal group's
    unsigned mb[6][2];
#pragma unroll
    for (int l = 0; l < 6; ++l) {
      mb[l][0] = mb[l][1] = 0u;
      if constexpr (TANGENT) {
        const u32x2v q = __builtin_nontemporal_load(reinterpret_cast<const u32x2v*>(A.bprim_bits) + ((size_t)l * A.bng_prim + gprim) * 64 + lane);
        mb[l][0] = q.x; mb[l][1] = q.y;
      }
    }
    constexpr int MODE = TANGENT ? EP_MASK : EP_RELU;
    // ---- trunk: 6 x Dense(128) + ReLU, skip concat [h, inputs] at layer 4 (warping.py:264-269) ----
    layer128<5, false, MODE, MODE, STASH, 0, 20, WF_SLOT>(c, acc0, acc1, ua, ua, mb[0][1], mb[0][0], hst(0), hst(0), WF_T, lane16,
        [&](int r) __attribute__((always_inline)) { return r == 0 ? bias_op : BF_ROWS(win, 1); });                              // L0: win -> ua
    layer128<9, true, MODE, MODE, STASH, 0, 36, WF_SLOT>(c, acc0, acc1, ua, ub, mb[0][1], mb[1][0], hst(0), hst(1), WF_T, lane16,
        [&](int r) __attribute__((always_inline)) { return r == 0 ? bias_op : BF_ROWS(ua, 1); });
    layer128<9, true, MODE, MODE, STASH, 0, 36, WF_SLOT>(c, acc0, acc1, ub, ua, mb[1][1], mb[2][0], hst(1), hst(2), WF_S, lane16,
        [&](int r) __attribute__((always_inline)) { return r == 0 ? bias_op : BF_ROWS(ub, 1); });
    layer128<9, true, MODE, MODE, STASH, 0, 36, WF_SLOT>(c, acc0, acc1, ua, ub, mb[2][1], mb[3][0], hst(2), hst(3), WF_T5, lane16,
        [&](int r) __attribute__((always_inline)) { return r == 0 ? bias_op : BF_ROWS(ua, 1); });
    layer128<13, true, MODE, MODE, STASH, 0, 52, WF_SLOT>(c, acc0, acc1, ub, ua, mb[3][1], mb[4][0], hst(3), hst(4), WF_L0, lane16,  // skip: [h, inputs]
        [&](int r) __attribute__((always_inline)) { return r == 0 ? bias_op : r <= 8 ? BF_ROWS(ub, 1) : BF_ROWS(win, 9); });
    layer128<9, true, MODE, MODE, STASH, 0, 45, WF_SLOT>(c, acc0, acc1, ua, ub, mb[4][1], mb[5][0], hst(4), hst(5), WF_T, lane16,
        [&](int r) __attribute__((always_inline)) { return r == 0 ? bias_op : BF_ROWS(ua, 1); });
    // ---- heads: w = Dense(128 -> 3)(h6), v = Dense(128 -> 3)(h6) (warping.py:271-288, 328-329): one block, features 0..2 = w,
    //      3..5 = v, the last 9 fragments of layer 5's chunk (the chunk's barrier falls here: the copy of layer 1 starts while the
    //      next iteration's layer 0 is already resident); h6's blocks 2, 3 (pending) are this panel's k-steps 4..7 ----
    f32x16 hd[1];
    {
      const __amdgpu_buffer_rsrc_t r51 = panel_rsrc(hst(5), 1);
      bf_panel<1, 9, true, 4, ep_ops(MODE, STASH), STASH, 36, 45, WF_SLOT, (STASH ? 8 : 0)>(hd, c.fr, c.rg, c.ll, wave, WF_T,
          [&](int r) __attribute__((always_inline)) { return r == 0 ? bias_op : BF_ROWS(ub, 1); },
          [&](int k) __attribute__((always_inline)) { wpanel_epi<4, 2, MODE, STASH>(k, acc1, ub, mb[5][1], r51, lane16); });
    }
    if constexpr (STASH && !TANGENT) {
#pragma unroll
      for (int l = 0; l < 6; ++l)
        __builtin_nontemporal_store((u32x2v){mb[l][0], mb[l][1]}, reinterpret_cast<u32x2v*>(A.bst.bits) + ((size_t)l * A.bst.ngroups + gidx) * 64 + lane);
    }
    // lane (n, 0): registers 0..3 = (w0, w1, w2, v0); lane (n, 1): registers 0, 1 = (v1, v2)
    const float v1 = __shfl_xor(hd[0][0], 32), v2 = __shfl_xor(hd[0][1], 32);
    if (h == 0 && row < A.rows) {
      const V3 w = v3(hd[0][0], hd[0][1], hd[0][2]), v = v3(hd[0][3], v1, v2);
      if (TANGENT) {   // (dw / dx_c, dv / dx_c): exp_se3's part of the Jacobian is applied by elastic_kernel / jacobian_kernel
        const size_t tr = (size_t)cdir * A.rows_pad + row;
        A.st_wv[2 * tr] = make_float4(w.x, w.y, w.z, 0.f);
        A.st_wv[2 * tr + 1] = make_float4(v.x, v.y, v.z, 0.f);
      } else {
        const V3 xw = se3_apply(w, v, v3(x[0], x[1], x[2]));
        float* o = A.points_out + (size_t)row * 3;
        o[0] = xw.x; o[1] = xw.y; o[2] = xw.z;
        if (STASH) {
          A.st_wv[2 * (size_t)row] = make_float4(w.x, w.y, w.z, 0.f);
          A.st_wv[2 * (size_t)row + 1] = make_float4(v.x, v.y, v.z, 0.f);
        }
      }
    }
  }
}

void launch_warp_fwd_bf16(const WarpFwdArgs& a, const WarpFwdArgs* a1, bool stash, int max_grid, hipStream_t stream) {
  const size_t lds = 3 * WF_SLOT;
  WarpBf16FwdArgs2 p;
  p.a[0] = a; p.a[1] = a1 ? *a1 : a;
  const bool tangent = a.bprim_bits != nullptr;
  p.n0 = (tangent ? 3 : 1) * ((a.rows + 255) / 256);
  p.ntot = p.n0 + (a1 ? (a1->rows + 255) / 256 : 0);
  const int grid = p.ntot < max_grid ? p.ntot : max_grid;
  const void* fn = tangent ? (const void*)se3_fwd_bf16_kernel<true, true>
                           : stash ? (const void*)se3_fwd_bf16_kernel<true, false> : (const void*)se3_fwd_bf16_kernel<false, false>;
  (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (tangent) hipLaunchKernelGGL((se3_fwd_bf16_kernel<true, true>), dim3(grid), dim3(512), lds, stream, p);
  else if (stash) hipLaunchKernelGGL((se3_fwd_bf16_kernel<true, false>), dim3(grid), dim3(512), lds, stream, p);
  else hipLaunchKernelGGL((se3_fwd_bf16_kernel<false, false>), dim3(grid), dim3(512), lds, stream, p);
}

// ---------------------------------------------------------------------------------------------
// reverse (data gradients of the trunk, GLO-code gradient); bias and weight gradients come from the dY stash (wgrad_bf16.hip)
// ---------------------------------------------------------------------------------------------
// Iterations [0, n0) level 0 (coarse samples), [n0, n01) level 1 (fine), [n01, ntot) level 2 (background points): ONE launch for
// the reverse passes through the shared field.  TANGENT: reverse of the tangent pass, starting from dL/d(dw/dx_c), dL/d(dv/dx_c)
// (written by elastic_kernel), masks of the primal groups, no code gradient (the tangent input does not depend on the code).
struct WarpBf16BwdArgs3 { WarpBwdArgs a[3]; int n0, n01, ntot; };
template <bool TANGENT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void se3_bwd_bf16_kernel(const WarpBf16BwdArgs3 P) {
  extern __shared__ __attribute__((aligned(16))) char bf_lds[];
  const int lane0 = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

  ChainCtx c;
  chain_start(c, bf_lds, P.a[0].bwpk, TANGENT ? WB_TAN_TOTAL : WB_TOTAL, WB_G5, WB_L, lane0, wave);

#pragma unroll 1
  for (int it = blockIdx.x; it < P.ntot; it += gridDim.x) {
    const int lv = it < P.n0 ? 0 : it < P.n01 ? 1 : 2;
    const WarpBwdArgs& A = P.a[lv];
    int lo = lane0;
    asm volatile("" : "+v"(lo));
    const int lane = lo, n = lane & 31, h = lane >> 5;
    const int lane16 = lane * 16;
    const int itl = it - (lv == 0 ? 0 : lv == 1 ? P.n0 : P.n01);
    const int nit_prim = (A.rows + 255) / 256;            // tangent: A.rows = the PRIMAL rows
    const int cdir = TANGENT ? itl / nit_prim : 0;
    const int pit = TANGENT ? itl - cdir * nit_prim : itl;
    const int row = pit * 256 + wave * 32 + n;
    const size_t gprim = (size_t)pit * 8 + wave;
    const size_t gidx = TANGENT ? (size_t)cdir * A.bng_prim + gprim : gprim;
    const BfWarpStash& S = A.bst;
    // ---- dL/d(w, v) of this row: exp_se3's VJP (primal), or the elastic kernel's tangent adjoints; both lane halves compute it ----
    V3 dw = v3(0.f, 0.f, 0.f), dv = dw;
    if (row < A.rows) {
      if (TANGENT) {
        const size_t tr = (size_t)cdir * A.rows_pad + row;
        const float4 a = A.d_w4[tr], b = A.d_v4[tr];
        dw = v3(a.x, a.y, a.z); dv = v3(b.x, b.y, b.z);
      } else {
        const V3 xx = v3(A.x_rows[3 * (size_t)row], A.x_rows[3 * (size_t)row + 1], A.x_rows[3 * (size_t)row + 2]);
        const float4 w4 = A.st_wv[2 * (size_t)row], v4 = A.st_wv[2 * (size_t)row + 1];
        const V3 g = v3(A.d_points[3 * (size_t)row], A.d_points[3 * (size_t)row + 1], A.d_points[3 * (size_t)row + 2]);
        se3_vjp<float>(v3(w4.x, w4.y, w4.z), v3(v4.x, v4.y, v4.z), xx, g, dw, dv);
        if (A.extra_dw4) {   // + the elastic regulariser's gradient w.r.t. the primal head outputs
          const float4 a = A.extra_dw4[row], b = A.extra_dv4[row];
          dw = dw + v3(a.x, a.y, a.z); dv = dv + v3(b.x, b.y, b.z);
        }
      }
    }
    // the "small" dY block: features 0..5 = (dw, dv); lane (n, 0) holds features 0..3, lane (n, 1) features 4, 5
    unsigned dsm[4];
    dsm[0] = h == 0 ? pack_bf16(dw.x, dw.y) : pack_bf16(dv.y, dv.z);
    dsm[1] = h == 0 ? pack_bf16(dw.z, dv.x) : 0u;
    dsm[2] = dsm[3] = 0u;
    {
      const __amdgpu_buffer_rsrc_t rsm = panel_rsrc(S.dhead + gidx * 2 * BF_BLOCK_DW, 0);
      bf_store16(rsm, lane16, dsm[0], dsm[1], 0u, 0u);
#pragma unroll
      for (int i = 1; i < 4; ++i) bf_store16(rsm, lane16 + i * BF_KB, 0u, 0u, 0u, 0u);
    }
    auto bits_of = [&](int l) __attribute__((always_inline)) {
      const uint32_t* base = TANGENT ? A.bprim_bits : S.bits;
      const int ng = TANGENT ? A.bng_prim : S.ngroups;
      return __builtin_nontemporal_load(reinterpret_cast<const u32x2v*>(base) + ((size_t)l * ng + gprim) * 64 + lane);
    };
    auto dyst = [&](int l) __attribute__((always_inline)) { return S.dy + ((size_t)l * S.ngroups + gidx) * 4 * BF_BLOCK_DW; };

    unsigned ua[4][8], ub[4][8];
    f32x16 acc0[2], acc1[2];
    // ---- heads^T: d h6 = [dw | dv] . [Ww | Wv]^T (K = 6 of one k-step + a zero one), mask of layer 5 -> dpre_5 = ua; one panel of
    //      4 blocks, its epilogue runs behind it (the next layer's first k-step already needs block 0) ----
    unsigned mq0, mq1;   // bit words (panel 0, panel 1) of the layer whose mask the current epilogues apply
    { const u32x2v q = bits_of(5); mq0 = q.x; mq1 = q.y; }
    {
      f32x16 g4[4];
      bf_panel<4, 2, true, 0, 0, false, 0, 40, BF_SLOT, 0>(g4, c.fr, c.rg, c.ll, wave, WB_L,     // the first 8 fragments of layer 5's chunk
          [&](int) __attribute__((always_inline)) { return as_bf16x8(dsm[0], dsm[1], dsm[2], dsm[3]); },
          [&](int) __attribute__((always_inline)) {});
      const __amdgpu_buffer_rsrc_t r0 = panel_rsrc(dyst(5), 0), r1 = panel_rsrc(dyst(5), 1);
#pragma unroll
      for (int k = 1; k <= 16; ++k) wpanel_epi<16, 0, EP_MASK, true, 0>(k, g4, ua, mq0, r0, lane16);
#pragma unroll
      for (int k = 1; k <= 16; ++k) wpanel_epi<16, 2, EP_MASK, true, 2>(k, g4, ua, mq1, r1, lane16);
    }
    // ---- l = 5..1: d h_l = W_l[0:128] . dpre_l, mask of layer l-1 -> dpre_{l-1}; the arrays alternate.  dpre_4 is kept for
    //      the code gradient (the skip layer sees the input) ----
    unsigned d4[4][8];
    // chunks two ahead: [heads^T + L5] L4 L3 L2 L1 [code GEMMs] (primal) -- behind L2 / L1 the stream wraps in the tangent pass
#define WB_LAYER(L, PEND, IN, OUT, F0, NFC, PRE, B2)                                                                           \
    {                                                                                                                       \
      unsigned mp1 = mq1;                                                                                                   \
      { const u32x2v q = bits_of((L) - 1); mq0 = q.x; mq1 = q.y; }                                                          \
      layer128<8, PEND, EP_MASK, EP_MASK, true, F0, NFC, BF_SLOT, PRE>(c, acc0, acc1, IN, OUT, mp1, mq0, dyst(L), dyst((L) - 1), B2, lane16, \
          [&](int r) __attribute__((always_inline)) { return BF_ROWS(IN, 0); });                                           \
    }
    WB_LAYER(5, false, ua, ub, 8, 40, 8, WB_L)    // -> dpre_4 = ub (blocks 2, 3 arrive during the next layer's first panel)
    WB_LAYER(4, true, ub, ua, 0, 32, 0, WB_L)     // -> dpre_3 = ua
    if constexpr (!TANGENT) {
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int q = 0; q < 8; ++q) d4[b][q] = ub[b][q];
    }
    WB_LAYER(3, true, ua, ub, 0, 32, 0, WB_L)                          // -> dpre_2 = ub
    WB_LAYER(2, true, ub, ua, 0, 32, 0, TANGENT ? WB_G5 : WB_L)        // -> dpre_1 = ua
    WB_LAYER(1, true, ua, ub, 0, 32, 0, TANGENT ? WB_L : WB_G5)        // -> dpre_0 = ub
#undef WB_LAYER
    if constexpr (TANGENT) {
      const __amdgpu_buffer_rsrc_t rp1 = panel_rsrc(dyst(0), 1);
#pragma unroll
      for (int k = 1; k <= 16; ++k) wpanel_epi<16, 2, EP_MASK, true>(k, acc1, ub, mq1, rp1, lane16);
    } else {
      // ---- GLO-code gradient: d input = W0 . dpre_0 + W4[128:] . dpre_4 (two 128 -> 64 GEMMs into one accumulator panel); the
      //      code columns of it, summed over the rows that share a warp id, are added to the embedding-table gradient ----
      f32x16 ci[2];
      const __amdgpu_buffer_rsrc_t rp1 = panel_rsrc(dyst(0), 1);
      bf_panel<2, 8, true, 6, ep_ops(EP_MASK, true), true, 0, 32, BF_SLOT, 0>(ci, c.fr, c.rg, c.ll, wave, WB_L,
          [&](int r) __attribute__((always_inline)) { return BF_ROWS(ub, 0); },
          [&](int k) __attribute__((always_inline)) { wpanel_epi<6, 2, EP_MASK, true>(k, acc1, ub, mq1, rp1, lane16); });
      bf_panel<2, 8, false, 0, 0, false, 16, 32, BF_SLOT, 4>(ci, c.fr, c.rg, c.ll, wave, WB_L,
          [&](int r) __attribute__((always_inline)) { return BF_ROWS(d4, 0); }, [&](int) __attribute__((always_inline)) {});
      int id = -1;
      if (row < A.rows) id = A.point_ids ? A.point_ids[row] : A.warp_ids ? A.warp_ids[row / A.S] : row / A.S;
      const int id0 = __builtin_amdgcn_readfirstlane(id);
      const bool one_id = __all(id == id0);   // a ray has >= 32 samples: the groups of the sample levels (padding rows break it)
      const int cbase = 3 + 6 * A.F;
      // one id per group: the (<= 8) column sums are collected in lanes 0 .. G-1 and leave as ONE atomic instruction = one request on the
      // id's 32-byte row (rounds 4-5: one instruction per column, 98 k requests per config-D launch onto the 64 lines of a 256-frame
      // table -- ~25 us of the kernel, profiles/r06_experiments.md section 4)
      float mine = 0.f;
#pragma unroll
      for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
          const int e0 = 32 * o + 8 * (rr >> 2) + (rr & 3);          // feature of this register in the h = 0 lanes (+ 4: h = 1)
          if (e0 + 4 < cbase || e0 >= cbase + A.G) continue;          // wave-uniform: neither half holds a code column
          const int g = e0 + 4 * h - cbase;
          const bool valid = g >= 0 && g < A.G && id >= 0;
          float v = valid ? ci[o][rr] : 0.f;
          if (one_id) {
#pragma unroll
            for (int sft = 16; sft > 0; sft >>= 1) v += __shfl_xor(v, sft);   // over the 32 rows of the lane half
            const float vo = __shfl_xor(v, 32);                                // the other half's column
            const int g0 = e0 - cbase, g1 = g0 + 4;                            // columns of the h = 0 / h = 1 lanes (wave-uniform)
            if (g0 >= 0 && g0 < A.G && lane == g0) mine = v;
            if (g1 >= 0 && g1 < A.G && lane == g1) mine = vo;
          } else if (valid && v != 0.f) {
            atomicAdd(A.grad_embed + (size_t)id * A.G + g, v);
          }
        }
      if (one_id && id0 >= 0 && lane < A.G && mine != 0.f) atomicAdd(A.grad_embed + (size_t)id0 * A.G + lane, mine);
    }
  }
}

void launch_warp_bwd_bf16(const WarpBwdArgs& a, const WarpBwdArgs* a1, const WarpBwdArgs* a2, int max_grid, hipStream_t stream) {
  const size_t lds = BF_LDS_BYTES;
  WarpBf16BwdArgs3 p;
  p.a[0] = a; p.a[1] = a1 ? *a1 : a; p.a[2] = a2 ? *a2 : a;
  p.n0 = (a.tangent ? 3 : 1) * ((a.rows + 255) / 256);
  p.n01 = p.n0 + (a1 ? (a1->rows + 255) / 256 : 0);
  p.ntot = p.n01 + (a2 ? (a2->rows + 255) / 256 : 0);
  const int grid = p.ntot < max_grid ? p.ntot : max_grid;
  if (a.tangent) {
    (void)hipFuncSetAttribute((const void*)se3_bwd_bf16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(se3_bwd_bf16_kernel<true>, dim3(grid), dim3(512), lds, stream, p);
  } else {
    (void)hipFuncSetAttribute((const void*)se3_bwd_bf16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(se3_bwd_bf16_kernel<false>, dim3(grid), dim3(512), lds, stream, p);
  }
}

}  // namespace nrf
