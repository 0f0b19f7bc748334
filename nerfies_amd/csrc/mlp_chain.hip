// Fused NeRF-MLP chain for gfx950: posenc -> 8x256 trunk (skip at 4) ->
// alpha head / bottleneck -> rgb branch, forward and data-gradient passes.
//
// Replaces (reference, /root/reference/nerfies):
//   modules.SinusoidalEncoder   modules.py:172-228  (fused into the tile prologue)
//   modules.MLP / NerfMLP       modules.py:26-62, 65-169
//   nn.sigmoid / sigma_activation  models.py:276-277
//
// Design (one workgroup = 4 waves = one 64-row tile, persistent over tiles, TWO
// workgroups resident per CU so one's epilogues hide under the other's MFMAs):
//   * activations of the tile live in LDS, feature-major  act[k][64 rows]
//     (XOR-swizzled 16-byte granules so the MFMA epilogue's ds_write_b128 is
//     bank-conflict free; the A-operand ds_read_b64 covers a whole row);
//   * every layer is  acc[64 x 64 per wave] += A(LDS) x B(weights), with
//     v_mfma_f32_32x32x2_f32 (exact fp32 == an fmaf chain).  Rows are
//     interleaved so that MFMA row-block rb holds tile rows p = 2*i + rb: one
//     ds_read_b64 then feeds the A operand of both row blocks;
//   * weights are pre-packed per layer in B-fragment order, so each lane
//     streams its B operands with one coalesced global_load_dwordx4 per 16
//     MFMAs straight from L2 -- no LDS traffic for weights;
//   * the training stash is written straight from the accumulator registers in
//     "fragment-native" order (coalesced 1 KiB per wave store); the dgrad pass
//     and the wgrad GEMM read it back in the same order.
#include <stdio.h>
#include <stdlib.h>

#include "chain_common.h"
#include "philox.h"

namespace nrf {

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigma_activation(float x, int kind) {
  if (kind == 1) {  // softplus, computed as jax.nn.softplus = logaddexp(x, 0)
    return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x)));
  }
  return relu(x);
}

// The arguments are read through a run-time index into the kernarg segment (always 0: gridDim.x < 2^24): hipcc then fetches
// each field with a scalar load where it is used instead of keeping the whole 700-byte struct in SGPRs for the lifetime of the
// kernel (round 2: 116 spilled SGPRs, parked in VGPR lanes of a kernel that is out of VGPRs).
struct ChainFwdArgs1 { ChainFwdArgs a[1]; };
template <bool STASH>
__global__ __launch_bounds__(256, 2) void nerf_mlp_fwd_kernel(const ChainFwdArgs1 P) {
  const ChainFwdArgs& A = P.a[blockIdx.x >> 24];
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* act = smem;                 // [256][64] swizzled
  float* pe = smem + ACT_FLOATS;     // [PK][64]; reused as scratch after the skip layer
  const int tid0 = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int part = wave;             // per-row (VALU) phases: 4 threads per tile row, one per wave
  const float* __restrict__ prm = A.params;
  const int PK = A.PK;
  const int PKS = (PK + 31) / 32 * 32;   // features per posenc stash tile (whole 32-feature blocks)
  const int nq_pe = PK / 16;
  // in-kernel timeline (scripts/timeline_mlp.py): compiled in only with -DNRF_TIMELINE_BUILD -- its counters and clock values
  // are live across the whole kernel, in a kernel that is out of registers
#ifdef NRF_TIMELINE_BUILD
  int stamp_i = 0;
  auto STAMP = [&]() {
    if (A.timeline && blockIdx.x == 0 && (tid0 & 63) == 0 && stamp_i < 64) A.timeline[wave * 64 + stamp_i] = clock64();
    ++stamp_i;
  };
  unsigned long long wg_t0 = 0, wg_w0 = 0;
  if (A.timeline && tid0 == 0) { wg_t0 = clock64(); wg_w0 = wall_clock64(); }
#else
  auto STAMP = [&]() {};
#endif
  int* tslot = reinterpret_cast<int*>(pe);   // free between tiles
  const TileIter ti = tile_iter(A.ntiles, A.k_old);
  for (int tile = A.tile_counter ? next_tile(A.tile_counter, tslot) : ti.first; tile < (A.tile_counter ? A.ntiles : ti.end);
       tile = A.tile_counter ? next_tile(A.tile_counter, tslot, tile) : tile + ti.step) {
    STAMP();   // tile start
    // the lane index is made opaque once per tile: everything derived from it (fragment addresses, row indices, mask shifts)
    // is then recomputed per tile instead of being hoisted out of the tile loop into registers that live -- i.e. spill -- across
    // the whole kernel (hipcc hoists ~50 such per-lane constants otherwise)
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int j = lane & 31, h = lane >> 5;
    const int p = lane;                // tile row handled in the per-row (VALU) phases
    // ---- prologue: sample point + SinusoidalEncoder (modules.py:213-228) ----
    {
      int r = tile * TILE_ROWS + p;
      r = r < A.rows ? r : A.rows - 1;
      float x[3];
      if (A.points) {
        x[0] = A.points[3 * r]; x[1] = A.points[3 * r + 1]; x[2] = A.points[3 * r + 2];
      } else {
        const int ray = r / A.S;
        const float z = A.zvals[r];
#pragma unroll
        for (int c = 0; c < 3; ++c)   // origins + z_vals * directions  (model_utils.py:72-73)
          x[c] = __fadd_rn(A.origins[3 * ray + c], __fmul_rn(z, A.directions[3 * ray + c]));
      }
      auto put = [&](int k, float v) { pe[k * TILE_ROWS + p] = v; };
      if (part == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) put(c, x[c]);
      } else if (part == 1) {
        for (int k = A.P; k < PK; ++k) put(k, 0.f);
      }
      const float half_pi = 1.57079632679489661923f;   // fp32(pi/2), modules.py:222
      for (int f = part; f < A.F; f += 4) {
        const float fr = (float)(1 << f);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float a = __fmul_rn(x[c], fr);
          put(3 + (2 * f) * 3 + c, sinf(a));
          put(3 + (2 * f + 1) * 3 + c, sinf(__fadd_rn(a, half_pi)));
        }
      }
    }
    __syncthreads();
    if (STASH) stash_tile_from_lds(pe, PK, PKS / 32, A.st_pe + (size_t)tile * PKS * TILE_ROWS, wave, lane);   // posenc stash, coalesced

    STAMP();   // prologue done
    f32x16 acc[2][2];
    const float4* wpk4 = reinterpret_cast<const float4*>(A.wpk);
    const size_t st_h_layer = (size_t)A.ntiles * FRAG_TILE_256;       // floats
    const int wv_soff = wave * 2 * 8 * 1024;                           // bytes: this wave's slice of a tile

    // ---- trunk: 8 x Dense(256)+ReLU, skip concat [h, posenc] at layer 4 (modules.py:41-50) ----
    const float4* wL0 = wpk4 + (A.pk.fwd_L[0] / 4) + wave * (PK / 4) * 64;
    WQuad<2> wnext = prefetch_quad<2>(wL0, lane);
    BiasRegs<2> bnext = bias_load<2>(prm + A.po.trunk_b[0], wave * 64, lane);
#pragma unroll 1
    for (int l = 0; l < TRUNK_DEPTH; ++l) {
      bias_set<2>(acc, bnext);
      if (l == 0) {
        mfma_k_loop<2, false>(acc, pe, nq_pe, wL0, lane, wnext);
      } else {
        mfma_k_loop<2, true>(acc, act, 16, wpk4 + (A.pk.fwd_L[l] / 4) + wave * 64 * 64, lane, wnext);
        if (l == A.skip) {
          const float4* w4b = wpk4 + (A.pk.fwd_L4b / 4) + wave * (PK / 4) * 64;
          mfma_k_loop<2, false>(acc, pe, nq_pe, w4b, lane, prefetch_quad<2>(w4b, lane));
        }
      }
      // the next layer's first weights go out before this layer's stash stores
      wnext = prefetch_quad<2>(wpk4 + ((l + 1 < TRUNK_DEPTH ? A.pk.fwd_L[l + 1] : A.pk.fwd_bn) / 4) + wave * 64 * 64, lane);
      bnext = bias_load<2>(prm + (l + 1 < TRUNK_DEPTH ? A.po.trunk_b[l + 1] : A.po.bn_b), wave * 64, lane);   // ... and its bias (chain_common.h)
      __builtin_amdgcn_sched_barrier(0);
      STAMP();   // k loop of layer l issued
      fwd_epilogue<2, EPI_RELU, STASH>(
          acc, wave * 64, act,
          make_rsrc(STASH ? A.st_h + l * st_h_layer + (size_t)tile * FRAG_TILE_256 : nullptr, FRAG_TILE_256 * 4), wv_soff,
          STASH ? A.bits_trunk + (((size_t)l * A.ntiles + tile) * 4 + wave) * 128 : nullptr, lane);
      STAMP();   // epilogue of layer l done
    }

    // ---- alpha head: Dense(256->1) on the trunk output, or -- use_alpha_condition -- Dense(256+A->1) on
    //      [bottleneck, appearance code] with the per-ray code term from ray_prep (modules.py:152-157) ----
    float sigma_raw = 0.f;
    auto alpha_head = [&]() {
      // weights in chunks of 16 (wave-uniform -> one s_load_dwordx16 per chunk instead of a scalar load and a
      // wait per k), activations as 16 independent LDS reads
      const float4* __restrict__ wa4 = reinterpret_cast<const float4*>(prm + A.po.alpha_k) + part * 16;
      float s = 0.f;
      const int k0 = part * 64;
#pragma unroll 1
      for (int kc = 0; kc < 4; ++kc) {
        float4 w4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w4[i] = wa4[4 * kc + i];
        float a[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = act[act_elem(k0 + 16 * kc + i, p)];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          s = fmaf(a[4 * i], w4[i].x, s); s = fmaf(a[4 * i + 1], w4[i].y, s);
          s = fmaf(a[4 * i + 2], w4[i].z, s); s = fmaf(a[4 * i + 3], w4[i].w, s);
        }
      }
      pe[part * TILE_ROWS + p] = s;   // scratch (posenc no longer needed for this tile)
      __syncthreads();
      if (part == 0)
        sigma_raw = (pe[p] + pe[TILE_ROWS + p]) + (pe[2 * TILE_ROWS + p] + pe[3 * TILE_ROWS + p]) + prm[A.po.alpha_b];
    };
    if (!A.alpha_ct) alpha_head();

    STAMP();   // alpha head done
    // ---- bottleneck: Dense(256), no activation (modules.py:149-150) ----
    bias_set<2>(acc, bnext);
    mfma_k_loop<2, true>(acc, act, 16, wpk4 + (A.pk.fwd_bn / 4) + wave * 64 * 64, lane, wnext);
    const float4* wrgb = wpk4 + (A.pk.fwd_rgbh / 4) + wave * 32 * 64;
    const WQuad<1> wrgb0 = prefetch_quad<1>(wrgb, lane);
    __builtin_amdgcn_sched_barrier(0);
    fwd_epilogue<2, EPI_LINEAR, STASH>(
        acc, wave * 64, act,
        make_rsrc(STASH ? A.st_bn + (size_t)tile * FRAG_TILE_256 : nullptr, FRAG_TILE_256 * 4), wv_soff, nullptr, lane);
    if (A.alpha_ct) {
      alpha_head();   // the scratch is next written by the rgb logits, two barriers further on
      if (part == 0) sigma_raw += A.alpha_ct[min((tile * TILE_ROWS + p) / A.S, A.B - 1)];
    }

    STAMP();   // bottleneck done
    // ---- rgb branch hidden: Dense(256+R -> 128)+ReLU; the R per-ray condition columns are
    //      folded into condterm[ray][n] (= cond . W[256:] + bias) by ray_prep ----
    {
      f32x16 acc1[2][1];
      zero_acc<1>(acc1);
      const int n = wave * 32 + j;
      // rows visited by this lane increase with q: walk the ray boundaries instead of dividing.  The first
      // condition term is fetched before the K loop so that its latency hides under the MFMAs.
      int ray = (tile * TILE_ROWS) / A.S;
      int nextb = (ray + 1) * A.S - tile * TILE_ROWS;   // first tile row of the next ray
      float ct = A.condterm[(size_t)min(ray, A.B - 1) * RGB_W + n];
      mfma_k_loop<1, true>(acc1, act, 16, wrgb, lane, wrgb0);
      const __amdgpu_buffer_rsrc_t st = make_rsrc(STASH ? A.st_rgbh + (size_t)tile * FRAG_TILE_128 : nullptr, FRAG_TILE_128 * 4);
      __syncthreads();
      uint32_t mb = 0u;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int g = q_granule(q, h);
        const float4 a4 = acc_piece<1>(acc1, 0, q);
        const float av[4] = {a4.x, a4.y, a4.z, a4.w};
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int pr = 4 * g + e;
          while (pr >= nextb) { ++ray; nextb += A.S; ct = A.condterm[(size_t)min(ray, A.B - 1) * RGB_W + n]; }
          v[e] = av[e] + ct;
        }
        float4 v4 = make_float4(v[0], v[1], v[2], v[3]);
        if (STASH) mb |= sign_nibble(v4) << (4 * q);
        v4.x = relu(v4.x); v4.y = relu(v4.y); v4.z = relu(v4.z); v4.w = relu(v4.w);
        *reinterpret_cast<float4*>(act + act_addr(n, g)) = v4;
        if (STASH) buf_store4(v4, st, lane * 16, (wave * 8 + q) * 1024);
      }
      if (STASH) A.bits_rgbh[((size_t)tile * 4 + wave) * 64 + lane] = mb;
      __syncthreads();
    }

    STAMP();   // rgb hidden done
    // ---- rgb logits Dense(128->3), sigmoid; sigma activation (models.py:276-277) ----
    {
      // [128][3] row-major: this part's 32 k = 96 consecutive floats, read as 6 chunks of 16
      const float4* __restrict__ wl4 = reinterpret_cast<const float4*>(prm + A.po.logit_k) + part * 24;
      float sc[3] = {0.f, 0.f, 0.f};
      const int k0 = part * 32;
#pragma unroll 1
      for (int kc = 0; kc < 2; ++kc) {   // 16 k = 48 weights per trip
        float4 w4[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) w4[i] = wl4[12 * kc + i];
        const float* wf = reinterpret_cast<const float*>(w4);
        float a[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = act[act_elem(k0 + 16 * kc + i, p)];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          sc[0] = fmaf(a[i], wf[3 * i], sc[0]); sc[1] = fmaf(a[i], wf[3 * i + 1], sc[1]); sc[2] = fmaf(a[i], wf[3 * i + 2], sc[2]);
        }
      }
      const float s0 = sc[0], s1 = sc[1], s2 = sc[2];
      pe[(3 * part) * TILE_ROWS + p] = s0; pe[(3 * part + 1) * TILE_ROWS + p] = s1; pe[(3 * part + 2) * TILE_ROWS + p] = s2;
      __syncthreads();
      if (part == 0) {
        float t[3];
#pragma unroll
        for (int c = 0; c < 3; ++c)
          t[c] = (pe[c * TILE_ROWS + p] + pe[(3 + c) * TILE_ROWS + p]) + (pe[(6 + c) * TILE_ROWS + p] + pe[(9 + c) * TILE_ROWS + p]) +
                 prm[A.po.logit_b + c];
        float4 o;
        o.x = 1.f / (1.f + expf(-t[0])); o.y = 1.f / (1.f + expf(-t[1])); o.z = 1.f / (1.f + expf(-t[2]));
        if (A.noise_std > 0.f) {   // model_utils.noise_regularize (model_utils.py:266-282)
          const int row = tile * TILE_ROWS + p;
          const float nz = A.noise ? A.noise[min(row, A.rows - 1)]
                                   : philox_normal(A.dyn ? A.dyn->rng_seed : A.noise_seed, A.dyn ? A.dyn->rng_offset : A.noise_offset, A.noise_stream, (uint32_t)row);
          sigma_raw = __fadd_rn(sigma_raw, __fmul_rn(nz, A.noise_std));
        }
        o.w = sigma_activation(sigma_raw, A.sigma_act);
        A.out4[(size_t)tile * TILE_ROWS + p] = o;
      }
      __syncthreads();   // scratch (aliases pe) is free again for the next tile's prologue
    }
  }
#ifdef NRF_TIMELINE_BUILD
  if (A.timeline && tid0 == 0) {   // per-workgroup residency record: start, end (shader clock), HW_ID, XCC_ID
    unsigned long long* rec = A.timeline + 1024 + 4 * (size_t)blockIdx.x;
    rec[0] = wg_t0; rec[1] = clock64();
    rec[2] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID
    rec[3] = (__builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)) & 0xf)   // HW_REG_XCC_ID
             | ((wall_clock64() - wg_w0) << 8);                               // residency in 100 MHz ticks
  }
#endif
}

void launch_chain_fwd(const ChainFwdArgs& a, bool stash, int grid, hipStream_t stream) {
  const size_t lds = (size_t)(ACT_FLOATS + a.PK * TILE_ROWS) * sizeof(float);
  ChainFwdArgs1 p;
  p.a[0] = a;
  if (knobs().debug_occ) {
    int nb = -1;
    (void)hipFuncSetAttribute((const void*)nerf_mlp_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)nerf_mlp_fwd_kernel<true>, 256, lds);
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, (const void*)nerf_mlp_fwd_kernel<true>);
    fprintf(stderr, "[nrf] fwd<true>: lds %zu B, occupancy %d blocks/CU (err %d), regs %d, static lds %zu, scratch %zu\n", lds, nb, (int)e,
            fa.numRegs, fa.sharedSizeBytes, fa.localSizeBytes);
  }
  if (stash) {
    (void)hipFuncSetAttribute((const void*)nerf_mlp_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(nerf_mlp_fwd_kernel<true>, dim3(grid), dim3(256), lds, stream, p);
  } else {
    (void)hipFuncSetAttribute((const void*)nerf_mlp_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(nerf_mlp_fwd_kernel<false>, dim3(grid), dim3(256), lds, stream, p);
  }
}

// ---------------------------------------------------------------------------------------------
// backward (data gradients; bias gradients accumulated per workgroup)
// ---------------------------------------------------------------------------------------------
// small_part layout (floats): db_trunk[8][256] | db_bn[256] | db_rgbh[128] | db_logit[3] | db_alpha
constexpr int SP_DB_TRUNK = 0, SP_DB_BN = 2048, SP_DB_RGBH = 2304, SP_DB_LOGIT = 2432, SP_DB_ALPHA = 2435;

// per-lane bias-gradient accumulators of one workgroup, carried across its tiles of one level
struct BwdAcc {
  float db_trunk[TRUNK_DEPTH][2];
  float db_bn[2];
  float db_rgbh;
  float dsum[4];   // threads < 64: column sums of d_raw (logit / alpha bias grads)
};
__device__ __forceinline__ void bwd_acc_zero(BwdAcc& c) {
#pragma unroll
  for (int l = 0; l < TRUNK_DEPTH; ++l) c.db_trunk[l][0] = c.db_trunk[l][1] = 0.f;
  c.db_bn[0] = c.db_bn[1] = 0.f;
  c.db_rgbh = 0.f;
  c.dsum[0] = c.dsum[1] = c.dsum[2] = c.dsum[3] = 0.f;
}

// one 64-row tile of level A (tile = index inside the level)
__device__ __forceinline__ void bwd_tile(const ChainBwdArgs& A, const int tile, float* smem, BwdAcc& C) {
  float* act = smem;                 // [256][64] swizzled: current dpre tile
  float* dr = smem + ACT_FLOATS;     // [4][64]: d raw rgb (3) and d raw sigma of the tile rows
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const float* __restrict__ prm = A.params;
  const float4* wpk4 = reinterpret_cast<const float4*>(A.wpk);
  const size_t layer_fl = (size_t)A.ntiles * FRAG_TILE_256;   // floats per trunk layer
  const int wv = wave * 2 * 8 * 1024;                           // bytes: this wave's slice of a tile
  float (&db_trunk)[TRUNK_DEPTH][2] = C.db_trunk;
  float (&db_bn)[2] = C.db_bn;
  float& db_rgbh = C.db_rgbh;
  float (&dsum)[4] = C.dsum;
  {
    if (tid < TILE_ROWS) {
      const float4 d = A.d_raw4[(size_t)tile * TILE_ROWS + tid];
      dr[tid] = d.x; dr[TILE_ROWS + tid] = d.y; dr[2 * TILE_ROWS + tid] = d.z; dr[3 * TILE_ROWS + tid] = d.w;
      dsum[0] += d.x; dsum[1] += d.y; dsum[2] += d.z; dsum[3] += d.w;
    }
    __syncthreads();

    // ---- rgb logit^T (3 -> 128) on the VALU, ReLU mask of the rgb hidden layer ----
    {
      int ln = lane;
      asm volatile("" : "+v"(ln));   // section-local lane constants (see the d posenc section)
      const int lane = ln, j = ln & 31, h = ln >> 5;
      const int n = wave * 32 + j;
      const float w0 = prm[A.po.logit_k + 3 * n], w1 = prm[A.po.logit_k + 3 * n + 1], w2 = prm[A.po.logit_k + 3 * n + 2];
      const uint32_t mb = A.bits_rgbh[((size_t)tile * 4 + wave) * 64 + lane];
      const __amdgpu_buffer_rsrc_t dy = make_rsrc(A.dy_rgbh + (size_t)tile * FRAG_TILE_128, FRAG_TILE_128 * 4);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int g = q_granule(q, h);
        const float4 d0 = *reinterpret_cast<const float4*>(dr + 4 * g);
        const float4 d1 = *reinterpret_cast<const float4*>(dr + TILE_ROWS + 4 * g);
        const float4 d2 = *reinterpret_cast<const float4*>(dr + 2 * TILE_ROWS + 4 * g);
        float4 v4 = make_float4(d0.x * w0 + d1.x * w1 + d2.x * w2, d0.y * w0 + d1.y * w1 + d2.y * w2,
                                d0.z * w0 + d1.z * w1 + d2.z * w2, d0.w * w0 + d1.w * w1 + d2.w * w2);
        v4 = mask4(v4, (mb >> (4 * q)) & 15u);
        db_rgbh += (v4.x + v4.y) + (v4.z + v4.w);
        *reinterpret_cast<float4*>(act + act_addr(n, g)) = v4;
        buf_store4(v4, dy, lane * 16, (wave * 8 + q) * 1024);
      }
    }
    __syncthreads();
    // ---- per-ray sums of dpre_rgbh (gradient of the per-ray condition columns of the rgb branch):
    //      thread (n, half) walks 32 tile rows of feature n in LDS and flushes at ray boundaries ----
    {
      int t2 = tid;
      asm volatile("" : "+v"(t2));
      const int n = t2 & 127, hf = t2 >> 7;
      const int row0 = 32 * hf;
      const int grow0 = tile * TILE_ROWS + row0;
      int ray = grow0 / A.S;
      int nextb = (ray + 1) * A.S - tile * TILE_ROWS;   // first tile row of the next ray
      const int nvalid = A.rows - tile * TILE_ROWS;      // tile rows >= nvalid are padding
      float ray_sum = 0.f;
#pragma unroll 1
      for (int g = row0 / 4; g < row0 / 4 + 8; ++g) {
        const float4 v4 = *reinterpret_cast<const float4*>(act + act_addr(n, g));
        const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int pr = 4 * g + e;
          while (pr >= nextb) {
            if (ray < A.B && ray_sum != 0.f) atomicAdd(A.dray + (size_t)ray * RGB_W + n, ray_sum);
            ray_sum = 0.f; ++ray; nextb += A.S;
          }
          if (pr < nvalid) ray_sum += v[e];
        }
      }
      if (ray < A.B && ray_sum != 0.f) atomicAdd(A.dray + (size_t)ray * RGB_W + n, ray_sum);
    }

    f32x16 acc[2][2];
    // ---- d bottleneck = dpre_rgbh . W_rgbh[0:256]^T   (K=128 -> N=256), linear ----
    zero_acc<2>(acc);
    {
      const float4* w0 = wpk4 + (A.pk.bwd_rgbhT / 4) + wave * 32 * 64;
      mfma_k_loop<2, true>(acc, act, 8, w0, lane, prefetch_quad<2>(w0, lane));
    }
    WQuad<2> wnext = prefetch_quad<2>(wpk4 + (A.pk.bwd_bnT / 4) + wave * 64 * 64, lane);
    __builtin_amdgcn_sched_barrier(0);
    {
      const __amdgpu_buffer_rsrc_t dy = make_rsrc(A.dy_bn + (size_t)tile * FRAG_TILE_256, FRAG_TILE_256 * 4);
      __syncthreads();
      int le = tid;
      asm volatile("" : "+v"(le));
      const int lane = le & 63, j = lane & 31, h = lane >> 5;
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const int n = wave * 64 + 32 * cb + j;
        const float wab = A.alpha_on_bn ? prm[A.po.alpha_k + n] : 0.f;   // use_alpha_condition: the alpha head reads the bottleneck
        float bsum = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          float4 v = acc_piece<2>(acc, cb, q);
          if (A.alpha_on_bn) {
            const float4 ds = *reinterpret_cast<const float4*>(dr + 3 * TILE_ROWS + 4 * q_granule(q, h));
            v.x = fmaf(ds.x, wab, v.x); v.y = fmaf(ds.y, wab, v.y); v.z = fmaf(ds.z, wab, v.z); v.w = fmaf(ds.w, wab, v.w);
          }
          bsum += (v.x + v.y) + (v.z + v.w);
          *reinterpret_cast<float4*>(act + act_addr(n, q_granule(q, h))) = v;
          buf_store4(v, dy, lane * 16, wv + (cb * 8 + q) * 1024);
        }
        db_bn[cb] += bsum;
      }
      __syncthreads();
    }

    // ---- d h8 = dbn . W_bn^T + d sigma_raw (x) w_alpha ; mask h8 > 0 -> dpre_7 ----
    // ---- then l = 7..1:  d h_l = dpre_l . W_l[0:256]^T ; mask h_l > 0 -> dpre_{l-1} ----
#pragma unroll 1
    for (int l = TRUNK_DEPTH; l >= 1; --l) {
      // the output of this step is dpre_{l-1}; its mask is sign(pre_{l-1}) = bits_trunk[l-1]
      const uint2 mq = *reinterpret_cast<const uint2*>(A.bits_trunk + (((size_t)(l - 1) * A.ntiles + tile) * 4 + wave) * 128 + lane * 2);
      const uint32_t mb[2] = {mq.x, mq.y};
      zero_acc<2>(acc);
      const int woff = (l == TRUNK_DEPTH) ? A.pk.bwd_bnT : A.pk.bwd_LT[l];
      mfma_k_loop<2, true>(acc, act, 16, wpk4 + (woff / 4) + wave * 64 * 64, lane, wnext);
      wnext = prefetch_quad<2>(wpk4 + (A.pk.bwd_LT[l > 1 ? l - 1 : 1] / 4) + wave * 64 * 64, lane);
      __builtin_amdgcn_sched_barrier(0);
      const __amdgpu_buffer_rsrc_t dy =
          make_rsrc(A.dy_trunk + (size_t)(l - 1) * layer_fl + (size_t)tile * FRAG_TILE_256, FRAG_TILE_256 * 4);
      __syncthreads();
      int le = tid;
      asm volatile("" : "+v"(le));   // epilogue-local lane constants: not live across the K loop
      const int lane = le & 63, j = lane & 31, h = lane >> 5;
      float bs[2];
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const int n = wave * 64 + 32 * cb + j;
        const float wa = (l == TRUNK_DEPTH && !A.alpha_on_bn) ? prm[A.po.alpha_k + n] : 0.f;
        float bsum = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int g = q_granule(q, h);
          float4 v = acc_piece<2>(acc, cb, q);
          if (l == TRUNK_DEPTH) {
            const float4 ds = *reinterpret_cast<const float4*>(dr + 3 * TILE_ROWS + 4 * g);
            v.x = fmaf(ds.x, wa, v.x); v.y = fmaf(ds.y, wa, v.y); v.z = fmaf(ds.z, wa, v.z); v.w = fmaf(ds.w, wa, v.w);
          }
          v = mask4(v, (mb[cb] >> (4 * q)) & 15u);
          bsum += (v.x + v.y) + (v.z + v.w);
          *reinterpret_cast<float4*>(act + act_addr(n, g)) = v;
          buf_store4(v, dy, lane * 16, wv + (cb * 8 + q) * 1024);
        }
        bs[cb] = bsum;
      }
      // runtime layer index -> static register: add into the matching accumulator
#pragma unroll
      for (int q = 0; q < TRUNK_DEPTH; ++q)
        if (q == l - 1) { db_trunk[q][0] += bs[0]; db_trunk[q][1] += bs[1]; }
      __syncthreads();

      // ---- warp on: d posenc = dpre_4 . W4[256:]^T + dpre_0 . W0^T  (256 -> PK columns).  Wave w owns
      //      MFMA row block w&1 (tile rows 2i + rb) x column block w>>1; the result goes to the dpe tile
      //      in LDS (aliases dr, dead since the l = 8 step), element (n, row) at n*64 + (row ^ (n & 31)). ----
      if (A.d_points && (l - 1 == A.skip || l == 1)) {
        float* dpe = dr;
        const bool first = (l - 1 == A.skip);
        int ln = lane;
        asm volatile("" : "+v"(ln));   // opaque: the per-lane constants of this section are recomputed here, not hoisted out of
                                       // the tile loop into registers that live (= spill) across the trunk layers
        const float4* wq = wpk4 + ((first ? A.pk.bwd_L4bT : A.pk.bwd_L0T) / 4) + ln;
        const int rb = wave & 1, cb = wave >> 1;
        f32x16 a2;
#pragma unroll
        for (int r = 0; r < 16; ++r) a2[r] = 0.f;
        const int j = ln & 31, h = ln >> 5;
        const int i = j, kk = h;
        const int aoff = 2 * (i & 1) + rb;
        const int PKS = (A.PK + 31) / 32 * 32;
        const int npw = PKS / 32 * 2;   // pieces per wave: 2 or 4
        // B from L2 in batches of 4 float4, the next batch in flight under the current one's MFMAs
        auto load_b = [&](float4 (&b)[4], int bt) {
#pragma unroll
          for (int u = 0; u < 4; ++u) b[u] = wq[(bt * 4 + u) * 64];
        };
        auto mma_b = [&](const float4 (&b)[4], int bt) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int k0 = 4 * (bt * 4 + u) + kk;
            const float a0 = act[act_addr(k0, i >> 1) + aoff];
            const float a1 = act[act_addr(k0 + 2, i >> 1) + aoff];
            a2 = mfma32(a0, cb ? b[u].y : b[u].x, a2);
            a2 = mfma32(a1, cb ? b[u].w : b[u].z, a2);
          }
        };
        float4 b0[4], b1[4];
        load_b(b0, 0);
#pragma unroll 1
        for (int bt = 0; bt < 16; bt += 2) {
          load_b(b1, bt + 1);
          mma_b(b0, bt);
          if (bt + 2 < 16) load_b(b0, bt + 2);
          mma_b(b1, bt + 1);
        }
        const int n = 32 * cb + j;
        if (n < A.PK) {
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) {
            const int row = 2 * c_row(reg, h) + rb;
            float* o = dpe + n * TILE_ROWS + (row ^ (n & 31));
            *o = first ? a2[reg] : *o + a2[reg];
          }
        }
        __syncthreads();
        if (!first) {
          // chain rule through SinusoidalEncoder: d sin(f x) = f cos(f x), d sin(f x + pi/2) = -f sin(f x),
          // with sin / cos taken from the forward posenc stash.  Every thread holds 4 rows of one feature per piece:
          // feature k's term  -+ 2^f * pe[k] * dpe[partner(k)]  goes to the contrib tile (act is dead), element (k, row)
          // at k*64 + (row ^ (k & 31)); 192 threads then sum their (row, c) over the 2F features in a fixed order.
          float* contrib = act;
          const int nfeat = 3 + 6 * A.F;
          float4 pv[4];
          {
            const float4* pe4 = reinterpret_cast<const float4*>(A.st_pe + (size_t)tile * PKS * TILE_ROWS) + ln;
#pragma unroll
            for (int u = 0; u < 4; ++u)
              if (u < npw) pv[u] = pe4[(wave + 4 * u) * 64];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (u < npw) {
              const int pid = wave + 4 * u, q = pid & 7;
              const int k = (pid >> 3) * 32 + j, g = (q & 1) + 2 * h + 4 * (q >> 1);
              if (k >= 3 && k < nfeat) {
                const int f = (k - 3) / 6, r = (k - 3) - 6 * f;
                const int partner = r < 3 ? k + 3 : k - 3;
                const float sgn = r < 3 ? -(float)(1 << f) : (float)(1 << f);
                const float* dp = dpe + partner * TILE_ROWS;
                float* co = contrib + k * TILE_ROWS;
                const float pvv[4] = {pv[u].x, pv[u].y, pv[u].z, pv[u].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const int row = 4 * g + e;
                  co[row ^ (k & 31)] = sgn * pvv[e] * dp[row ^ (partner & 31)];
                }
              }
            }
          }
          __syncthreads();
          if (tid < 3 * TILE_ROWS) {
            const int row = tid / 3, c = tid - 3 * row;
            float dx = dpe[c * TILE_ROWS + (row ^ c)];
            for (int f = 0; f < A.F; ++f) {
              const int ns = 3 + 6 * f + c, nc = ns + 3;
              dx += contrib[ns * TILE_ROWS + (row ^ (ns & 31))] + contrib[nc * TILE_ROWS + (row ^ (nc & 31))];
            }
            A.d_points[(size_t)tile * TILE_ROWS * 3 + tid] = dx;
          }
        }
        if (!first) __syncthreads();   // dr (aliased) is rewritten by the next tile
      }
    }
  }
}

// the per-workgroup partials of level A -> small_part[blockIdx.x]
__device__ __forceinline__ void bwd_flush(const ChainBwdArgs& A, float* smem, const BwdAcc& C) {
  float* dr = smem + ACT_FLOATS;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const float (&db_trunk)[TRUNK_DEPTH][2] = C.db_trunk;
  const float (&db_bn)[2] = C.db_bn;
  const float db_rgbh = C.db_rgbh;
  const float (&dsum)[4] = C.dsum;
  __syncthreads();   // dr may still be read by the tile just finished
  // ---- flush the per-workgroup partials ----
  float* sp = A.small_part + (size_t)blockIdx.x * SMALL_PART;
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const int n = wave * 64 + 32 * cb + j;
#pragma unroll
    for (int l = 0; l < TRUNK_DEPTH; ++l) {
      const float v = db_trunk[l][cb] + __shfl_xor(db_trunk[l][cb], 32);
      if (h == 0) sp[SP_DB_TRUNK + l * TRUNK_W + n] = v;
    }
    const float vb = db_bn[cb] + __shfl_xor(db_bn[cb], 32);
    if (h == 0) sp[SP_DB_BN + n] = vb;
  }
  {
    const int n = wave * 32 + j;
    const float vr = db_rgbh + __shfl_xor(db_rgbh, 32);
    if (h == 0) sp[SP_DB_RGBH + n] = vr;
  }
  __syncthreads();
  if (tid < TILE_ROWS) {
#pragma unroll
    for (int c = 0; c < 4; ++c) dr[c * TILE_ROWS + tid] = dsum[c];
  }
  __syncthreads();
  if (tid < 4) {
    float s = 0.f;
    for (int q = 0; q < TILE_ROWS; ++q) s += dr[tid * TILE_ROWS + q];
    sp[tid < 3 ? SP_DB_LOGIT + tid : SP_DB_ALPHA] = s;
  }
  __syncthreads();
}

// ONE launch for the coarse and the fine MLP (the two backward passes are independent: no gradient flows from the fine
// pass into the coarse MLP): global tiles [0, nt0) are level 0, [nt0, ntot) level 1, dealt round-robin, so a
// workgroup runs its 2 coarse tiles and goes straight on with its 6 fine ones (config A) instead of ramping up and draining
// twice.  The bias partials are flushed per level.  The level's arguments are indexed in the kernarg segment (scalar loads,
// one copy of the tile code).
struct ChainBwdArgs2 { ChainBwdArgs a[2]; int nt0, ntot; };
__global__ __launch_bounds__(256, 2) void nerf_mlp_bwd_kernel(const ChainBwdArgs2 P) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  BwdAcc C;
  bwd_acc_zero(C);
  const int nt0 = P.nt0, ntot = P.ntot;
  int cur = -1;
  if ((int)blockIdx.x >= nt0 && nt0 > 0) bwd_flush(P.a[0], smem, C);   // no coarse tile for this workgroup: its partial is zero
#pragma unroll 1
  for (int g = blockIdx.x; g < ntot; g += gridDim.x) {
    const int lv = g >= nt0 ? 1 : 0;
    if (lv != cur) {
      if (cur == 0) { bwd_flush(P.a[0], smem, C); bwd_acc_zero(C); }
      cur = lv;
    }
    bwd_tile(P.a[lv], g - (lv ? nt0 : 0), smem, C);
  }
  if (cur == 0) {
    bwd_flush(P.a[0], smem, C);
    if (ntot > nt0) { bwd_acc_zero(C); bwd_flush(P.a[1], smem, C); }
  } else if (cur == 1) {
    bwd_flush(P.a[1], smem, C);
  }
}

void launch_chain_bwd(const ChainBwdArgs& a0, const ChainBwdArgs* a1, int grid, hipStream_t stream) {
  const size_t lds = (size_t)(ACT_FLOATS + (a0.d_points ? a0.PK : 4) * TILE_ROWS) * sizeof(float);
  (void)hipFuncSetAttribute((const void*)nerf_mlp_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  ChainBwdArgs2 p;
  p.a[0] = a0; p.a[1] = a1 ? *a1 : a0;
  p.nt0 = a0.ntiles; p.ntot = p.nt0 + (a1 ? a1->ntiles : 0);
  hipLaunchKernelGGL(nerf_mlp_bwd_kernel, dim3(grid), dim3(256), lds, stream, p);
}

// ---------------------------------------------------------------------------------------------
// weight packing: canonical [in,out] kernels -> per-wave B-fragment streams
// ---------------------------------------------------------------------------------------------
__global__ void pack_weights_kernel(const PackDesc* __restrict__ descs, const float* __restrict__ params,
                                    float* __restrict__ ws) {
  const PackDesc d = descs[blockIdx.y];
  const float* __restrict__ src = params + d.src_off;
  float* __restrict__ dst = ws + d.dst_off;
  const int ncols_wave = d.ncb == 2 ? 64 : 32;
  const int total = d.K * ncols_wave * d.nwaves;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int e = idx & 3, lane = (idx >> 2) & 63;
    const int per_it = 256;                       // floats per (wave, it)
    const int nit = d.ncb == 2 ? d.K / 4 : d.K / 8;
    const int it = (idx / per_it) % nit, w = idx / (per_it * nit);
    int k, n;
    if (d.ncb == 2) { k = 4 * it + 2 * (e >> 1) + (lane >> 5); n = 64 * w + 32 * (e & 1) + (lane & 31); }
    else            { k = 8 * it + 2 * e + (lane >> 5);        n = 32 * w + (lane & 31); }
    float v = 0.f;
    if (k < d.kvalid && n < d.nvalid)
      v = d.transposed ? src[(size_t)(d.src_row0 + n) * d.src_ld + k] : src[(size_t)(d.src_row0 + k) * d.src_ld + n];
    dst[idx] = v;
  }
}

void launch_pack(const PackDesc* d_descs, int ndesc, const float* params, float* ws, hipStream_t stream) {
  hipLaunchKernelGGL(pack_weights_kernel, dim3(64, ndesc), dim3(256), 0, stream, d_descs, params, ws);
}

}  // namespace nrf
