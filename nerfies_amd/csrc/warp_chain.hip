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
#include "../../include/nerfies_amd.h"
#include "chain_common.h"
#include "general_loss.h"
#include "se3_math.h"

namespace nrf {

// small_part layout (floats): db_trunk[6][128] | db_w[3] | db_v[3]
constexpr int WSP_DB_TRUNK = 0, WSP_DB_W = 768, WSP_DB_V = 771;

// In-kernel timeline (scripts/timeline_warp.py), compiled in only with -DNRF_TIMELINE_BUILD: shader-clock stamps of the
// first tile of workgroup 0, per wave, of the LAST launch of each kernel flavour: [fwd primal, fwd tangent, bwd primal,
// bwd tangent][wave][stamp].
#ifdef NRF_TIMELINE_BUILD
__device__ unsigned long long g_warp_tl[4][4][64];
#define WSTAMP_INIT(K) int stamp_i_ = 0; const int stamp_k_ = (K); const bool stamp_on_ = blockIdx.x == 0 && tile == 0 && (threadIdx.x & 63) == 0 && A.S > 1   /* a sample level, not the background points */
#define WSTAMP() do { if (stamp_on_ && stamp_i_ < 64) g_warp_tl[stamp_k_][threadIdx.x >> 6][stamp_i_] = clock64(); ++stamp_i_; } while (0)
#else
#define WSTAMP_INIT(K)
#define WSTAMP()
#endif

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
// TANGENT: forward-mode pass of the warp Jacobian (warping.py:385-387): tile tt = c * nt_prim + t carries the
// tangent of primal tile t along coordinate c through the trunk (no biases, ReLU derivative = the primal
// sign bits) and emits (dw/dx_c, dv/dx_c) per row; exp_se3's part of the Jacobian is applied by elastic_kernel.
template <bool STASH, bool TANGENT>
__device__ __forceinline__ void warp_fwd_tile(const WarpFwdArgs& A, const int tile, float* smem) {
  float* act = smem;                  // [128][64] swizzled
  float* win = smem + WACT_FLOATS;    // [PKw][64] trunk input; reused as scratch after the skip layer
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));       // opaque per tile: per-lane constants are recomputed per tile, not hoisted out of the tile
                                      // loop into registers that live across the whole kernel (mlp_chain.hip, bwd_tile)
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane, part = wave;    // per-row phases: 4 threads per tile row
  const float* __restrict__ prm = A.params;
  const int PKw = A.PKw;
  const int PKS = (PKw + 31) / 32 * 32;
  const float4* wpk4 = reinterpret_cast<const float4*>(A.wpk);
  const size_t st_layer = (size_t)A.ntiles * FRAG_TILE_128;
  WSTAMP_INIT(TANGENT ? 1 : 0);
  WSTAMP();   // tile start
  {
    // ---- prologue: sample point, AnnealedSinusoidalEncoder (modules.py:231-294), GLO code ----
    float x[3] = {0.f, 0.f, 0.f};
    const int row = tile * TILE_ROWS + p;
    const int tprim = TANGENT ? tile % A.nt_prim : tile;   // primal tile whose masks / inputs this tile uses
    if (TANGENT) {
      // d input / d x_c from the primal input tile: d(win sin a) = f (win cos a), d(win cos a) = -f (win sin a)
      const int c = tile / A.nt_prim;
      const float* pw = A.prim_win + (size_t)tprim * PKS * TILE_ROWS;
      auto put = [&](int k, float v) { win[k * TILE_ROWS + p] = v; };
      if (part == 0) {
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) put(cc, cc == c ? 1.f : 0.f);
      } else if (part == 1) {
        for (int k = 3 + 6 * A.F; k < PKw; ++k) put(k, 0.f);
      }
      for (int f = part; f < A.F; f += 4) {
        const float fr = (float)(1 << f);
        const int ns = 3 + 6 * f, nc = ns + 3;
        const float sn = pw[frag_index(ns + c, p)], cs = pw[frag_index(nc + c, p)];
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
          put(ns + cc, cc == c ? fr * cs : 0.f);
          put(nc + cc, cc == c ? -fr * sn : 0.f);
        }
      }
    } else {
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
        id = A.warp_ids ? A.warp_ids[ray] : ray;   // nullptr: per-ray codes (metadata_encoded / TimeEncoder output)
      }
      auto put = [&](int k, float v) { win[k * TILE_ROWS + p] = v; };
      if (part == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) put(c, x[c]);
        if (A.points_raw && row < A.rows) {
          A.points_raw[3 * row] = x[0]; A.points_raw[3 * row + 1] = x[1]; A.points_raw[3 * row + 2] = x[2];
        }
      } else if (part == 1) {
        const float* __restrict__ code = A.embed_table + (int64_t)id * A.G;   // glo.py:50-53
        for (int g = 0; g < A.G; ++g) put(3 + 6 * A.F + g, code[g]);
        for (int k = A.Win; k < PKw; ++k) put(k, 0.f);
      }
      const float half_pi = 1.57079632679489661923f;
      const float pi = 3.14159265358979323846f;
      const float warp_alpha = A.dyn ? A.dyn->warp_alpha : A.alpha;   // device-resident in a graph-replayed step
      for (int f = part; f < A.F; f += 4) {
        // cosine_easing_window (modules.py:274-294): 0.5 (1 + cos(pi clip(alpha - band, 0, 1) + pi))
        const float cl = fminf(fmaxf(warp_alpha - (float)f, 0.f), 1.f);
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
    WSTAMP();   // prologue
    if (STASH) stash_tile_from_lds(win, PKw, PKS / 32, A.st_win + (size_t)tile * PKS * TILE_ROWS, wave, lane);   // trunk-input stash, coalesced

    // ---- trunk: 6 x Dense(128)+ReLU, skip concat [h, inputs] at layer 4 (warping.py:264-269) ----
    f32x16 acc[2][1];
    const int nq_in = PKw / 16, nit_in = PKw / 8;
    const float4* wL0 = wpk4 + (A.pk.fwd_L[0] / 4) + wave * nit_in * 64;
    WQuad<1> wnext = prefetch_quad<1>(wL0, lane);
    BiasRegs<1> bnext;
    if (!TANGENT) bnext = bias_load<1>(prm + A.po.trunk_b[0], wave * 32, lane);
#pragma unroll 1
    for (int l = 0; l < WARP_DEPTH; ++l) {
      if (TANGENT) zero_acc<1>(acc);
      else bias_set<1>(acc, bnext);
      if (l == 0) {
        mfma_k_loop<1, false>(acc, win, nq_in, wL0, lane, wnext);
      } else {
        mfma_k_loop<1, true>(acc, act, 8, wpk4 + (A.pk.fwd_L[l] / 4) + wave * 16 * 64, lane, wnext);
        if (l == WARP_SKIP) {
          const float4* w4b = wpk4 + (A.pk.fwd_L4b / 4) + wave * nit_in * 64;
          mfma_k_loop<1, false>(acc, win, nq_in, w4b, lane, prefetch_quad<1>(w4b, lane));
        }
      }
          WSTAMP();   // layer l: K loop
      wnext = prefetch_quad<1>(wpk4 + (A.pk.fwd_L[l + 1 < WARP_DEPTH ? l + 1 : l] / 4) + wave * 16 * 64, lane);
      if (!TANGENT) bnext = bias_load<1>(prm + A.po.trunk_b[l + 1 < WARP_DEPTH ? l + 1 : l], wave * 32, lane);   // before the stash stores
      __builtin_amdgcn_sched_barrier(0);
      if (TANGENT)
        fwd_epilogue<1, EPI_MASK, STASH>(
            acc, wave * 32, act,
            make_rsrc(STASH ? A.st_h + l * st_layer + (size_t)tile * FRAG_TILE_128 : nullptr, FRAG_TILE_128 * 4), wave * 8 * 1024,
            const_cast<uint32_t*>(A.prim_bits) + (((size_t)l * A.nt_prim + tprim) * 4 + wave) * 64, lane);
      else
        fwd_epilogue<1, EPI_RELU, STASH>(
            acc, wave * 32, act,
            make_rsrc(STASH ? A.st_h + l * st_layer + (size_t)tile * FRAG_TILE_128 : nullptr, FRAG_TILE_128 * 4),
            wave * 8 * 1024, STASH ? A.bits + (((size_t)l * A.ntiles + tile) * 4 + wave) * 64 : nullptr, lane);
      WSTAMP();   // layer l: epilogue
    }

    // ---- heads: w = Dense(128->3)(h), v = Dense(128->3)(h)  (warping.py:271-288, 328-329) ----
    {
      // on the MFMA pipe, K split over the four waves (mfma_kslice32): columns 0..2 = w, 3..5 = v; [128][3] leaves, element
      // (k, c) at 3k + c.  Lanes n < 6 hold this wave's partial sums of column n -> scratch [wave][column][row].
      float s[6];
      {
        const int n = lane & 31, hh = lane >> 5;
        const float* __restrict__ wsrc = prm + (n < 3 ? A.po.w_k + n : A.po.v_k + (n < 6 ? n - 3 : 0));
        f32x16 hacc[2];
        mfma_kslice32(hacc, act, 32 * part, lane, [&](int k) { return n < 6 ? wsrc[3 * k] : 0.f; });
        if (n < 6) {
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) {
            const int r0 = 2 * c_row(reg, hh);
            *reinterpret_cast<float2*>(win + (6 * part + n) * TILE_ROWS + r0) = make_float2(hacc[0][reg], hacc[1][reg]);
          }
        }
      }
      __syncthreads();
      WSTAMP();   // heads
      if (part == 0 && TANGENT) {
#pragma unroll
        for (int c = 0; c < 6; ++c)
          s[c] = (win[c * TILE_ROWS + p] + win[(6 + c) * TILE_ROWS + p]) + (win[(12 + c) * TILE_ROWS + p] + win[(18 + c) * TILE_ROWS + p]);
        A.st_wv[2 * (size_t)row] = make_float4(s[0], s[1], s[2], 0.f);       // dw / dx_c
        A.st_wv[2 * (size_t)row + 1] = make_float4(s[3], s[4], s[5], 0.f);   // dv / dx_c
      } else if (part == 0) {
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
      WSTAMP();   // exp_se3 + outputs
    }
  }
}

// Global tiles [0, nt0) belong to level 0, [nt0, ntot) to level 1 (dealt round-robin): the 256 tiles of the background-point
// batch (training.py:117-135) under-fill the chip on their own (warp_fwd_bg ran at 59 TF in round 2), so they ride in the
// launch of the coarse samples; same field, same packed weights.  The level's arguments are indexed in the kernarg segment
// (scalar loads; one copy of the tile code).
struct WarpFwdArgs2 { WarpFwdArgs a[2]; int nt0, ntot; };
template <bool STASH, bool TANGENT>
__global__ __launch_bounds__(256, NRF_WARP_WAVES) void se3_warp_fwd_kernel(const WarpFwdArgs2 P) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int nt0 = P.nt0, ntot = P.ntot;
#pragma unroll 1
  for (int g = blockIdx.x; g < ntot; g += gridDim.x) {
    const int lv = g >= nt0 ? 1 : 0;
    warp_fwd_tile<STASH, TANGENT>(P.a[lv], g - (lv ? nt0 : 0), smem);
  }
}

void launch_warp_fwd(const WarpFwdArgs& a, const WarpFwdArgs* a1, bool stash, int grid, hipStream_t stream) {
  const int pk = a.PKw < 32 ? 32 : a.PKw;   // the head scratch needs 24 rows
  const size_t lds = (size_t)(WACT_FLOATS + pk * TILE_ROWS) * sizeof(float);
  const void* fn = a.prim_win ? (const void*)se3_warp_fwd_kernel<true, true>
                              : stash ? (const void*)se3_warp_fwd_kernel<true, false> : (const void*)se3_warp_fwd_kernel<false, false>;
  (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  WarpFwdArgs2 p;
  p.a[0] = a; p.a[1] = a1 ? *a1 : a;
  p.nt0 = a.ntiles; p.ntot = p.nt0 + (a1 ? a1->ntiles : 0);
  if (a.prim_win) hipLaunchKernelGGL((se3_warp_fwd_kernel<true, true>), dim3(grid), dim3(256), lds, stream, p);
  else if (stash) hipLaunchKernelGGL((se3_warp_fwd_kernel<true, false>), dim3(grid), dim3(256), lds, stream, p);
  else hipLaunchKernelGGL((se3_warp_fwd_kernel<false, false>), dim3(grid), dim3(256), lds, stream, p);
}

// ---------------------------------------------------------------------------------------------
// backward (data gradients of the trunk, bias gradients, GLO-code gradient)
// ---------------------------------------------------------------------------------------------
// TANGENT: reverse of the tangent pass: starts from dL/d(dw/dx_c), dL/d(dv/dx_c) (written by elastic_kernel into
// d_w4 / d_v4), same masks as the primal tile, no bias / GLO-code gradients (the tangent input does not depend
// on them); its dY stash feeds the wgrad kernel together with the tangent activations.
struct WarpBwdAcc { float db[WARP_DEPTH]; float hsum[6]; };   // hsum: threads < 64, column sums of (dw, dv)

template <bool TANGENT>
__device__ __forceinline__ void warp_bwd_tile(const WarpBwdArgs& A, const int tile, float* smem, WarpBwdAcc& C) {
  float* act = smem;                       // [128][64] swizzled: current dpre tile
  float* dwv = smem + WACT_FLOATS;         // [8][64]: dL/dw (0..2), dL/dv (3..5) of the tile rows
  float* dcs = dwv + 8 * TILE_ROWS;        // [8][64]: dL/dcode of the tile rows
  int* ids_s = reinterpret_cast<int*>(dcs + 8 * TILE_ROWS);   // [64]: warp id of the tile rows (-1: padding)
  float* cgp = dcs + 9 * TILE_ROWS;        // [4 waves][8 codes][64]: K-slice partials of the GLO-code gradient
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));       // opaque per tile (see warp_fwd_tile)
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int p = lane, part = wave;
  const float* __restrict__ prm = A.params;
  const float4* wpk4 = reinterpret_cast<const float4*>(A.wpk);
  const size_t layer_fl = (size_t)A.ntiles * FRAG_TILE_128;
  const int PKS = (A.PKw + 31) / 32 * 32;
  const int n = wave * 32 + j;             // this lane's trunk column
  float (&db)[WARP_DEPTH] = C.db;
  float (&hsum)[6] = C.hsum;
  WSTAMP_INIT(TANGENT ? 3 : 2);
  WSTAMP();   // tile start
  {
    const int tprim = TANGENT ? tile % A.nt_prim : tile;
    // ---- exp_se3 VJP per row ----
    if (TANGENT) {
      if (tid < TILE_ROWS) {
        const int row = tile * TILE_ROWS + tid;
        const float4 a = A.d_w4[row], b = A.d_v4[row];
        dwv[tid] = a.x; dwv[TILE_ROWS + tid] = a.y; dwv[2 * TILE_ROWS + tid] = a.z;
        dwv[3 * TILE_ROWS + tid] = b.x; dwv[4 * TILE_ROWS + tid] = b.y; dwv[5 * TILE_ROWS + tid] = b.z;
      }
    } else if (tid < TILE_ROWS) {
      const int row = tile * TILE_ROWS + tid;
      V3 dw = v3(0.f, 0.f, 0.f), dv = dw;
      if (row < A.rows) {
        const float* sw = A.st_win + (size_t)tile * PKS * TILE_ROWS;
        const V3 x = v3(sw[frag_index(0, tid)], sw[frag_index(1, tid)], sw[frag_index(2, tid)]);
        const float4 w4 = A.st_wv[2 * (size_t)row], v4 = A.st_wv[2 * (size_t)row + 1];
        const V3 g = v3(A.d_points[3 * (size_t)row], A.d_points[3 * (size_t)row + 1], A.d_points[3 * (size_t)row + 2]);
        se3_vjp<float>(v3(w4.x, w4.y, w4.z), v3(v4.x, v4.y, v4.z), x, g, dw, dv);
        if (A.extra_dw4) {   // + the elastic regulariser's gradient w.r.t. the primal head outputs
          const float4 a = A.extra_dw4[row], b = A.extra_dv4[row];
          dw = dw + v3(a.x, a.y, a.z); dv = dv + v3(b.x, b.y, b.z);
        }
      }
      dwv[tid] = dw.x; dwv[TILE_ROWS + tid] = dw.y; dwv[2 * TILE_ROWS + tid] = dw.z;
      dwv[3 * TILE_ROWS + tid] = dv.x; dwv[4 * TILE_ROWS + tid] = dv.y; dwv[5 * TILE_ROWS + tid] = dv.z;
      A.d_w4[row] = make_float4(dw.x, dw.y, dw.z, 0.f);
      A.d_v4[row] = make_float4(dv.x, dv.y, dv.z, 0.f);
      hsum[0] += dw.x; hsum[1] += dw.y; hsum[2] += dw.z; hsum[3] += dv.x; hsum[4] += dv.y; hsum[5] += dv.z;
    }
    __syncthreads();
    WSTAMP();   // exp_se3 VJP

    // ---- heads^T (6 -> 128): d h5 = [dw | dv] . [Ww | Wv]^T as 4 MFMA k-steps (K = 6 padded to 8), ReLU mask of trunk
    //      layer 5 -> dpre_5 through the same epilogue as the trunk steps ----
    {
      const uint32_t mb = A.bits[(((size_t)(WARP_DEPTH - 1) * A.nt_prim + tprim) * 4 + wave) * 64 + lane];
      const __amdgpu_buffer_rsrc_t dy =
          make_rsrc(A.dy + (size_t)(WARP_DEPTH - 1) * layer_fl + (size_t)tile * FRAG_TILE_128, FRAG_TILE_128 * 4);
      f32x16 hacc[2][1];
      zero_acc<1>(hacc);
#pragma unroll
      for (int sk = 0; sk < 3; ++sk) {   // k = 2 sk + h < 6
        const int k = 2 * sk + h;
        const float b = prm[(k < 3 ? A.po.w_k + k : A.po.v_k + (k - 3)) + 3 * n];      // [128][3] leaves: element (n, c) at 3n + c
        const float2 a = *reinterpret_cast<const float2*>(dwv + k * TILE_ROWS + 2 * j);   // tile rows 2j, 2j+1 of component k
        hacc[0][0] = mfma32(a.x, b, hacc[0][0]);
        hacc[1][0] = mfma32(a.y, b, hacc[1][0]);
      }
      float bsum = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        float4 v4 = acc_piece<1>(hacc, 0, q);
        v4 = mask4(v4, (mb >> (4 * q)) & 15u);
        bsum += (v4.x + v4.y) + (v4.z + v4.w);
        *reinterpret_cast<float4*>(act + act_addr(n, q_granule(q, h))) = v4;
        buf_store4(v4, dy, lane * 16, (wave * 8 + q) * 1024);
      }
      db[WARP_DEPTH - 1] += bsum;
    }
    __syncthreads();
    WSTAMP();   // heads^T

    // GLO-code gradient: d code[g] = dpre_l . W_l[row_base + g][:]^T for the two layers that see the
    // input (l = 4 via the skip rows, l = 0), K = 128 on the VALU; thread = (row p, codes 2*part, 2*part+1).
    auto code_grad = [&](int64_t krow_off, bool first) {
      // on the MFMA pipe, K split over the four waves (mfma_kslice32): B[k][g] = W_l[row_base + g][k]; lanes n < 8 hold this
      // wave's partial of code n and keep it in their own LDS slot cgp[wave][code][row] (the layer-0 call adds onto the
      // layer-4 call's values: same lanes, same slot, no barrier); the four partials are summed once, in the scatter below
      int lo = lane;
      asm volatile("" : "+v"(lo));   // section-local lane constants
      const int nn = lo & 31, hh = lo >> 5;
      const float* __restrict__ wsrc = prm + krow_off + (int64_t)min(nn, A.G - 1) * WARP_W;
      f32x16 cacc[2];
      mfma_kslice32(cacc, act, 32 * wave, lo, [&](int k) { return nn < A.G ? wsrc[k] : 0.f; });
      if (nn < 8) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          float2* slot = reinterpret_cast<float2*>(cgp + (wave * 8 + nn) * TILE_ROWS + 2 * c_row(reg, hh));
          float2 v = make_float2(cacc[0][reg], cacc[1][reg]);
          if (!first) { const float2 o = *slot; v.x += o.x; v.y += o.y; }
          *slot = v;
        }
      }
    };

    // ---- l = 5..1: d h_l = dpre_l . W_l[0:128]^T ; mask h_l > 0 -> dpre_{l-1} ----
    f32x16 acc[2][1];
    WQuad<1> wnext = prefetch_quad<1>(wpk4 + (A.pk.bwd_LT[WARP_DEPTH - 1] / 4) + wave * 16 * 64, lane);
#pragma unroll 1
    for (int l = WARP_DEPTH - 1; l >= 1; --l) {
      if (!TANGENT && l == WARP_SKIP) { code_grad(A.po.trunk_k[WARP_SKIP] + (int64_t)(WARP_W + 3 + 6 * A.F) * WARP_W, true); WSTAMP(); }
      const uint32_t mb = A.bits[(((size_t)(l - 1) * A.nt_prim + tprim) * 4 + wave) * 64 + lane];
      zero_acc<1>(acc);
      mfma_k_loop<1, true>(acc, act, 8, wpk4 + (A.pk.bwd_LT[l] / 4) + wave * 16 * 64, lane, wnext);
          WSTAMP();   // step l: K loop
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
      WSTAMP();   // step l: epilogue
    }
    if (TANGENT) return;
    code_grad(A.po.trunk_k[0] + (int64_t)(3 + 6 * A.F) * WARP_W, false);
    WSTAMP();   // code gradient (layer 0 rows)

    // ---- sums of d code over the rows of the tile that share a warp id -> scatter-add into the embedding-table
    //      gradient.  One atomic per (distinct id of the tile, code): the background batch carries a random id per point
    //      (training.py:121-123), where summing runs of equal ids sent 64 x G atomics per tile to a table of a few rows
    //      (same-address device atomics serialise at ~12 ns: round 2's warp_dgrad_bg spent more time there than in its
    //      MFMAs).  Thread (g, q): if row q is the first of the tile with its id, it owns that id's sum. ----
    // every wave looks at the ids of all 64 rows (lane = row): the waves agree on the path without a barrier
    int id;
    {
      const int grow = tile * TILE_ROWS + lane;
      id = -1;
      if (grow < A.rows) id = A.point_ids ? A.point_ids[grow] : A.warp_ids ? A.warp_ids[grow / A.S] : grow / A.S;
    }
    const int id0 = __shfl(id, 0);   // row 0 of a tile is never padding
    const bool one_id = __all(id == id0 || id < 0);
    __syncthreads();   // cgp complete
    if (one_id) {
      // all valid rows share one id (a ray has >= 64 samples: the sample levels): wave w owns codes 2w, 2w+1 -- sum of the four
      // K-slice partials per row, one shuffle reduction over the rows, one atomic
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int g = 2 * wave + q;
        float v = 0.f;
        if (id >= 0 && g < A.G)
          v = (cgp[(0 * 8 + g) * TILE_ROWS + lane] + cgp[(1 * 8 + g) * TILE_ROWS + lane]) +
              (cgp[(2 * 8 + g) * TILE_ROWS + lane] + cgp[(3 * 8 + g) * TILE_ROWS + lane]);
        const float sm = wave_sum_f(v);
        if (lane == 0 && g < A.G && sm != 0.f) atomicAdd(A.grad_embed + (size_t)id0 * A.G + g, sm);
      }
    } else {
      // per-point ids (the background batch): thread (g, q) -- if row q is the first of the tile with its id, it owns that
      // id's sum
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int g = 2 * part + q;
        dcs[g * TILE_ROWS + p] = (cgp[(0 * 8 + g) * TILE_ROWS + p] + cgp[(1 * 8 + g) * TILE_ROWS + p]) +
                                 (cgp[(2 * 8 + g) * TILE_ROWS + p] + cgp[(3 * 8 + g) * TILE_ROWS + p]);
      }
      if (wave == 0) ids_s[lane] = id;
      __syncthreads();
      const int g = tid & 7;
#pragma unroll 1
      for (int half = 0; half < 2; ++half) {
        const int q = (tid >> 3) + 32 * half;
        const int idq = ids_s[q];
        if (g < A.G && idq >= 0) {
          bool leader = true;
          for (int e = 0; e < q; ++e) leader = leader && ids_s[e] != idq;
          if (leader) {
            float sm = 0.f;
            for (int e = q; e < TILE_ROWS; ++e) sm += ids_s[e] == idq ? dcs[g * TILE_ROWS + e] : 0.f;
            if (sm != 0.f) atomicAdd(A.grad_embed + (size_t)idq * A.G + g, sm);
          }
        }
      }
    }
    __syncthreads();
    WSTAMP();   // embedding-gradient scatter
  }
}

// Global tiles [0, n0) are level 0 (coarse samples), [n0, n01) level 1 (fine samples), [n01, ntot) level 2 (background
// points): ONE launch for the three reverse passes through the shared field (round 2: three launches at 72 / 83 / 24 TF).
// The bias partials of all levels add up in the workgroup's registers and are flushed once (the leaves are shared).
struct WarpBwdArgs3 { WarpBwdArgs a[3]; int n0, n01, ntot; };
template <bool TANGENT>
__global__ __launch_bounds__(256, NRF_WARP_WAVES) void se3_warp_bwd_kernel(const WarpBwdArgs3 P) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* dwv = smem + WACT_FLOATS;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int n = wave * 32 + j;
  const int n0 = P.n0, n01 = P.n01, ntot = P.ntot;
  WarpBwdAcc C;
#pragma unroll
  for (int l = 0; l < WARP_DEPTH; ++l) C.db[l] = 0.f;
#pragma unroll
  for (int c = 0; c < 6; ++c) C.hsum[c] = 0.f;
#pragma unroll 1
  for (int g = blockIdx.x; g < ntot; g += gridDim.x) {
    const int lv = g < n0 ? 0 : g < n01 ? 1 : 2;
    warp_bwd_tile<TANGENT>(P.a[lv], g - (lv == 0 ? 0 : lv == 1 ? n0 : n01), smem, C);
  }
  if (TANGENT) return;
  const WarpBwdArgs& A = P.a[0];
  const float (&db)[WARP_DEPTH] = C.db;
  const float (&hsum)[6] = C.hsum;
  __syncthreads();
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

void launch_warp_bwd(const WarpBwdArgs& a, const WarpBwdArgs* a1, const WarpBwdArgs* a2, int grid, hipStream_t stream) {
  const size_t lds = (size_t)(WACT_FLOATS + (17 + 32) * TILE_ROWS) * sizeof(float);
  WarpBwdArgs3 p;
  p.a[0] = a; p.a[1] = a1 ? *a1 : a; p.a[2] = a2 ? *a2 : a;
  p.n0 = a.ntiles; p.n01 = p.n0 + (a1 ? a1->ntiles : 0); p.ntot = p.n01 + (a2 ? a2->ntiles : 0);
  if (a.tangent) {
    (void)hipFuncSetAttribute((const void*)se3_warp_bwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(se3_warp_bwd_kernel<true>, dim3(grid), dim3(256), lds, stream, p);
  } else {
    (void)hipFuncSetAttribute((const void*)se3_warp_bwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(se3_warp_bwd_kernel<false>, dim3(grid), dim3(256), lds, stream, p);
  }
}

// ---------------------------------------------------------------------------------------------
// elastic regulariser (training.compute_elastic_loss, training.py:71-114, 177-197; loss_type 'log_svals')
// ---------------------------------------------------------------------------------------------
// Symmetric 3x3 eigen-decomposition (cyclic Jacobi): D = V diag(mu) V^T.
// Four sweeps: on 2e5 random J^T J - I of every scale the off-diagonal is <= 5e-12 |D| after the fourth (3e-5 after the third;
// rounds 1-5 ran six, the last two on a diagonal matrix).  The rotation ANGLE only steers the convergence -- any (c, s) with
// c^2 + s^2 = 1 is an exact similarity -- so theta and t = tan come from the hardware reciprocal / square root (1 ulp, no
// IEEE division sequences: 3 of the 4 per rotation), and only c = rsqrt(1 + t^2) is refined to float accuracy (one Newton step).
__device__ __forceinline__ void jacobi3(float (&D)[3][3], float (&V)[3][3]) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int k = 0; k < 3; ++k) V[i][k] = i == k ? 1.f : 0.f;
#pragma unroll 1
  for (int sweep = 0; sweep < 4; ++sweep) {
#pragma unroll
    for (int pq = 0; pq < 3; ++pq) {
      const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
      const float apq = D[p][q];
      const bool live = fabsf(apq) >= 1e-30f;
      const float theta = (D[q][q] - D[p][p]) * __builtin_amdgcn_rcpf(2.f * (live ? apq : 1.f));
      const float t = copysignf(__builtin_amdgcn_rcpf(fabsf(theta) + __builtin_amdgcn_sqrtf(fmaf(theta, theta, 1.f))), theta);
      const float u = fmaf(t, t, 1.f);
      float c = __builtin_amdgcn_rsqf(u);
      c = c * fmaf(-0.5f * u, c * c, 1.5f);          // Newton: c <- c (3 - u c^2) / 2
      c = live ? c : 1.f;
      const float s = live ? t * c : 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) {   // D <- D J
        const float dkp = D[k][p], dkq = D[k][q];
        D[k][p] = c * dkp - s * dkq; D[k][q] = s * dkp + c * dkq;
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {   // D <- J^T D
        const float dpk = D[p][k], dqk = D[q][k];
        D[p][k] = c * dpk - s * dqk; D[q][k] = s * dpk + c * dqk;
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float vkp = V[k][p], vkq = V[k][q];
        V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq;
      }
    }
  }
}

// E = J - I of one sample: column c = d/dx_c [exp_se3(w, v) x - x], from the raw head outputs (w, v), their tangents
// (wd_c, vd_c) along x_c and the point x (Dual evaluation of se3_delta).
__device__ __forceinline__ void warp_jacobian_minus_identity(const Se3CoefD& kc, V3 w, V3 v, V3 x, const V3 (&wd)[3], const V3 (&vd)[3],
                                                             float (&E)[3][3]) {
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const V3T<Dual> W = v3t<Dual>(Dual(w.x, wd[c].x), Dual(w.y, wd[c].y), Dual(w.z, wd[c].z));
    const V3T<Dual> Vv = v3t<Dual>(Dual(v.x, vd[c].x), Dual(v.y, vd[c].y), Dual(v.z, vd[c].z));
    const V3T<Dual> X = v3t<Dual>(Dual(x.x, c == 0 ? 1.f : 0.f), Dual(x.y, c == 1 ? 1.f : 0.f), Dual(x.z, c == 2 ? 1.f : 0.f));
    const V3T<Dual> dl = se3_delta_c<Dual>(se3_coef_along(kc, 2.f * dot(w, wd[c])), W, Vv, X);   // the coefficients: once per sample (se3_math.h)
    E[0][c] = dl.x.d; E[1][c] = dl.y.d; E[2][c] = dl.z.d;
  }
}

// One thread per coarse sample.  E = J - I (above);
// J^T J - I = E + E^T + E^T E = V diag(mu) V^T;  log s_k = 0.5 log1p(mu_k)  (accurate near the identity);
// sq = the squared residual of elastic_loss_type (training.py:86-109: 'log_svals' sum log(max(s_k, eps))^2, 'svals'
// sum (s_k - 1)^2, 'jtj' |J J^T - I|^2 / 4, 'div' tr(E)^2, 'det' (det J - 1)^2, 'log_det' log(max(det J, eps))^2);
// rho = general_loss(sq, alpha, scale);  L = (1/B) sum_rows coef_row rho_row.
// dL/dJ = coef/B * weight * rho'(sq) * d sq/dJ  (singular-value types: J V diag((d sq/d s_k) / s_k) V^T);  its pull-back
// through exp_se3 comes from se3_vjp on Duals: value parts -> adjoints of (wd_c, vd_c), tangent parts (Hessian-vector
// products) -> adjoints of the primal (w, v).  Also the Jacobian statistics of training.py:214-222.
__global__ __launch_bounds__(256) void elastic_kernel(const ElasticArgs A) {
  // Per-thread 3x3 scratch (column = threadIdx.x: no two threads share a word, no barrier): E, later gs * d sq/dJ.  The three
  // Dual evaluations of each phase run as a ROLLED loop over the column c -- a third of the straight-line code (the kernel
  // executes every instruction once per wave, i.e. it streams its own text through the instruction cache), no 36 registers
  // of (wd, vd, wdb, vdb) arrays; the tangents are re-read from L2 in the second loop instead.
  __shared__ float e_s[9][256];
  const int tid = threadIdx.x;
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  float rho_c = 0.f, res = 0.f, jdet = 0.f, jdiv = 0.f, jcurl = 0.f;
  if (row < A.rows_pad) {
    V3 wbar = v3(0.f, 0.f, 0.f), vbar = wbar;
    if (row < A.rows) {
      V3 x;
      if (A.x_rows) {   // bf16 trunk: the points are kept as plain fp32 rows
        x = v3(A.x_rows[3 * (size_t)row], A.x_rows[3 * (size_t)row + 1], A.x_rows[3 * (size_t)row + 2]);
      } else {
        const int tile = row / TILE_ROWS, p = row % TILE_ROWS;
        const float* sw = A.prim_win + (size_t)tile * A.PKS * TILE_ROWS;
        x = v3(sw[frag_index(0, p)], sw[frag_index(1, p)], sw[frag_index(2, p)]);
      }
      const float4 w4 = A.prim_wv[2 * (size_t)row], v4 = A.prim_wv[2 * (size_t)row + 1];
      const V3 w = v3(w4.x, w4.y, w4.z), v = v3(v4.x, v4.y, v4.z);
      const Se3CoefD kc = se3_coef_d(dot(w, w));   // shared by the three Jacobian columns and the three pull-backs below
      float E[3][3];
#pragma unroll 1
      for (int c = 0; c < 3; ++c) {   // E = J - I, column c = d/dx_c [exp_se3(w, v) x - x]  (Dual evaluation of se3_delta)
        const size_t tr = (size_t)c * A.rows_pad + row;
        const float4 a = A.tan_wv[2 * tr], b = A.tan_wv[2 * tr + 1];
        const V3T<Dual> W = v3t<Dual>(Dual(w.x, a.x), Dual(w.y, a.y), Dual(w.z, a.z));
        const V3T<Dual> Vv = v3t<Dual>(Dual(v.x, b.x), Dual(v.y, b.y), Dual(v.z, b.z));
        const V3T<Dual> X = v3t<Dual>(Dual(x.x, c == 0 ? 1.f : 0.f), Dual(x.y, c == 1 ? 1.f : 0.f), Dual(x.z, c == 2 ? 1.f : 0.f));
        const V3T<Dual> dl = se3_delta_c<Dual>(se3_coef_along(kc, 2.f * (w.x * a.x + w.y * a.y + w.z * a.z)), W, Vv, X);
        e_s[c][tid] = dl.x.d; e_s[3 + c][tid] = dl.y.d; e_s[6 + c][tid] = dl.z.d;
      }
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) E[i][k] = e_s[3 * i + k][tid];
      // det J - 1 = tr E + (principal 2x2 minors of E) + det E: no cancellation near the identity
      const float trE = E[0][0] + E[1][1] + E[2][2];
      const float m2 = (E[0][0] * E[1][1] - E[0][1] * E[1][0]) + (E[0][0] * E[2][2] - E[0][2] * E[2][0]) + (E[1][1] * E[2][2] - E[1][2] * E[2][1]);
      const float detE = E[0][0] * (E[1][1] * E[2][2] - E[1][2] * E[2][1]) - E[0][1] * (E[1][0] * E[2][2] - E[1][2] * E[2][0]) +
                         E[0][2] * (E[1][0] * E[2][1] - E[1][1] * E[2][0]);
      const float dm1 = trE + m2 + detE;
      jdet = 1.f + dm1; jdiv = trE;                                   // utils.jacobian_to_div (utils.py:85-91)
      const float c0 = E[2][1] - E[1][2], c1 = E[0][2] - E[2][0], c2 = E[1][0] - E[0][1];   // utils.jacobian_to_curl (:71-84)
      jcurl = sqrtf(c0 * c0 + c1 * c1 + c2 * c2);
      float sq = 0.f;
      float Gd[3][3];   // d sq / dJ
      if (A.loss_type == NRF_ELASTIC_DIV) {
        sq = trE * trE;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int k = 0; k < 3; ++k) Gd[i][k] = i == k ? 2.f * trE : 0.f;
      } else if (A.loss_type == NRF_ELASTIC_DET || A.loss_type == NRF_ELASTIC_LOG_DET) {
        float J[3][3], Cf[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int k = 0; k < 3; ++k) J[i][k] = E[i][k] + (i == k ? 1.f : 0.f);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int k = 0; k < 3; ++k) {   // cofactor matrix: d det / dJ
            const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, k1 = (k + 1) % 3, k2 = (k + 2) % 3;
            Cf[i][k] = J[i1][k1] * J[i2][k2] - J[i1][k2] * J[i2][k1];
          }
        float f;
        if (A.loss_type == NRF_ELASTIC_DET) { sq = dm1 * dm1; f = 2.f * dm1; }
        else {
          const bool live = jdet > A.eps;
          const float ld = live ? log1pf(dm1) : logf(A.eps);
          sq = ld * ld; f = live ? 2.f * ld / jdet : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int k = 0; k < 3; ++k) Gd[i][k] = f * Cf[i][k];
      } else {
        float D[3][3], M[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int k = 0; k < 3; ++k) D[i][k] = E[i][k] + E[k][i] + (E[0][i] * E[0][k] + E[1][i] * E[1][k] + E[2][i] * E[2][k]);
        if (A.loss_type == NRF_ELASTIC_JTJ) {   // |J J^T - I|_F^2 / 4 = |J^T J - I|_F^2 / 4 ; d/dJ = J (J^T J - I)
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int k = 0; k < 3; ++k) { sq += 0.25f * D[i][k] * D[i][k]; M[i][k] = D[i][k]; }
        } else {
          float Vm[3][3], m[3];
          jacobi3(D, Vm);
          const float log_eps = logf(A.eps);
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const float mu = D[k][k], lam = 1.f + mu;
            if (A.loss_type == NRF_ELASTIC_SVALS) {
              const float sk = sqrtf(fmaxf(lam, 0.f)), sm1 = mu / (sk + 1.f);   // s - 1 without cancellation
              sq += sm1 * sm1;
              m[k] = sk > 1e-20f ? 2.f * sm1 / sk : 0.f;
            } else {
              const bool live = lam > A.eps * A.eps;     // s_k > eps (training.py:88)
              const float ls = live ? 0.5f * log1pf(mu) : log_eps;
              sq += ls * ls;
              m[k] = live ? 2.f * ls / lam : 0.f;         // (d sq / d s_k) / s_k
            }
          }
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int k = 0; k < 3; ++k) M[i][k] = Vm[i][0] * m[0] * Vm[k][0] + Vm[i][1] * m[1] * Vm[k][1] + Vm[i][2] * m[2] * Vm[k][2];
        }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int k = 0; k < 3; ++k) Gd[i][k] = M[i][k] + E[i][0] * M[0][k] + E[i][1] * M[1][k] + E[i][2] * M[2][k];   // J M, J = I + E
      }
      float rho, drho;
      general_loss_sq(sq, A.alpha, A.scale, rho, drho);   // utils.py:264-331, every branch
      const float coef = A.coef[row];
      rho_c = coef * rho;
      res = (A.res_selected && coef == 0.f) ? 0.f : sqrtf(sq);
      const float gs = coef * (A.dyn ? A.dyn->elastic_loss_weight * A.inv_rays : A.gscale) * drho;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) e_s[3 * i + k][tid] = gs * Gd[i][k];
#pragma unroll 1
      for (int c = 0; c < 3; ++c) {
        const size_t tr = (size_t)c * A.rows_pad + row;
        const float4 a = A.tan_wv[2 * tr], b = A.tan_wv[2 * tr + 1];
        const V3T<Dual> W = v3t<Dual>(Dual(w.x, a.x), Dual(w.y, a.y), Dual(w.z, a.z));
        const V3T<Dual> Vv = v3t<Dual>(Dual(v.x, b.x), Dual(v.y, b.y), Dual(v.z, b.z));
        const V3T<Dual> X = v3t<Dual>(Dual(x.x, c == 0 ? 1.f : 0.f), Dual(x.y, c == 1 ? 1.f : 0.f), Dual(x.z, c == 2 ? 1.f : 0.f));
        const V3T<Dual> g = v3t<Dual>(Dual(e_s[c][tid]), Dual(e_s[3 + c][tid]), Dual(e_s[6 + c][tid]));
        V3T<Dual> dw, dv;
        se3_vjp_c<Dual>(se3_coef_along(kc, 2.f * (w.x * a.x + w.y * a.y + w.z * a.z)), W, Vv, X, g, dw, dv);
        A.tan_dw4[tr] = make_float4(dw.x.v, dw.y.v, dw.z.v, 0.f);
        A.tan_dv4[tr] = make_float4(dv.x.v, dv.y.v, dv.z.v, 0.f);
        wbar = wbar + v3(dw.x.d, dw.y.d, dw.z.d); vbar = vbar + v3(dv.x.d, dv.y.d, dv.z.d);
      }
    } else {   // padding rows of the last 256-row iteration: the reverse tangent pass reads them
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const size_t tr = (size_t)c * A.rows_pad + row;
        A.tan_dw4[tr] = make_float4(0.f, 0.f, 0.f, 0.f);
        A.tan_dv4[tr] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    A.prim_dw4[row] = make_float4(wbar.x, wbar.y, wbar.z, 0.f);
    A.prim_dv4[row] = make_float4(vbar.x, vbar.y, vbar.z, 0.f);
  }
  // loss / residual / Jacobian-statistic sums: one partial per workgroup and quantity, summed by finish_stats_kernel in a fixed order.
  // (Rounds 1-5 added them with one atomic per wave and quantity: 2048 waves x 5 atomics on ONE 128-byte line serialise in the L2 --
  // 124 of the kernel's 142 us at the config-D shape, scripts/micro/elastic_bench.hip; without them 18 us.)
  float sm[5] = {rho_c, res, jdet, jdiv, jcurl};
#pragma unroll
  for (int q = 0; q < 5; ++q) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sm[q] += __shfl_xor(sm[q], o);
  }
  __syncthreads();                       // e_s is free: every thread is done with its column
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int q = 0; q < 5; ++q) e_s[q][threadIdx.x >> 6] = sm[q];
  }
  __syncthreads();
  if (threadIdx.x < 5)
    A.part[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = (e_s[threadIdx.x][0] + e_s[threadIdx.x][1]) + (e_s[threadIdx.x][2] + e_s[threadIdx.x][3]);
}

// return_warp_jacobian (models.py:264-265, warping.py:385-387): J = I + E per sample, row-major [3][3].
__global__ __launch_bounds__(256) void jacobian_kernel(const JacobianArgs A) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= A.rows) return;
  V3 x;
  if (A.x_rows) {
    x = v3(A.x_rows[3 * (size_t)row], A.x_rows[3 * (size_t)row + 1], A.x_rows[3 * (size_t)row + 2]);
  } else {
    const int tile = row / TILE_ROWS, p = row % TILE_ROWS;
    const float* sw = A.prim_win + (size_t)tile * A.PKS * TILE_ROWS;
    x = v3(sw[frag_index(0, p)], sw[frag_index(1, p)], sw[frag_index(2, p)]);
  }
  const float4 w4 = A.prim_wv[2 * (size_t)row], v4 = A.prim_wv[2 * (size_t)row + 1];
  V3 wd[3], vd[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const size_t tr = (size_t)c * A.rows_pad + row;
    const float4 a = A.tan_wv[2 * tr], b = A.tan_wv[2 * tr + 1];
    wd[c] = v3(a.x, a.y, a.z); vd[c] = v3(b.x, b.y, b.z);
  }
  float E[3][3];
  const V3 wj = v3(w4.x, w4.y, w4.z);
  warp_jacobian_minus_identity(se3_coef_d(dot(wj, wj)), wj, v3(v4.x, v4.y, v4.z), x, wd, vd, E);
  float* o = A.out + (size_t)row * 9;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int k = 0; k < 3; ++k) o[3 * i + k] = E[i][k] + (i == k ? 1.f : 0.f);
}

void launch_jacobian(const JacobianArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL(jacobian_kernel, dim3((a.rows + 255) / 256), dim3(256), 0, stream, a);
}

void launch_elastic(const ElasticArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL(elastic_kernel, dim3((a.rows_pad + 255) / 256), dim3(256), 0, stream, a);
}

}  // namespace nrf

#ifdef NRF_TIMELINE_BUILD
extern "C" __attribute__((visibility("default"))) int nrf_debug_warp_timeline(unsigned long long* host_dst) {
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(nrf::g_warp_tl), sizeof(nrf::g_warp_tl));
}
#endif
