// The fused NeRF-MLP chain of mlp_chain.hip on 32-ROW tiles, FOUR workgroups per CU (round 5).
//
// Same math, same packed weights, same HBM images (fragment-order stash, ReLU sign bits, d raw / out4 rows) as the 64-row
// kernels -- a 32-row workgroup owns one HALF of a 64-row stash tile, so wgrad, the reduce passes and every reader of the
// workspace are untouched and the two tilings can be mixed launch by launch (forward 32 / reverse 64 and so on).
//
// Replaces (reference, /root/reference/nerfies):
//   modules.SinusoidalEncoder   modules.py:172-228  (tile prologue)
//   modules.MLP / NerfMLP       modules.py:26-62, 65-169
//   nn.sigmoid / sigma_activation  models.py:276-277
//
// Why a second tiling.  The 64-row kernel keeps 128 accumulators + two weight sets per wave (256 VGPRs): two waves per SIMD.
// A single wave issues one fp32 MFMA per 68.8 clocks (64 is the pipe rate) and ~12 % of a tile is outside K loops (prologue,
// VALU heads, epilogues); with two waves per SIMD a pair of tiles co-runs at 88 % of the MFMA rate.  Here a wave owns 32 rows
// x 64 columns: 32 accumulators, two 16-register weight sets, 8 A registers -> <= 128 VGPRs, FOUR waves per SIMD (4 x 40 KiB
// of LDS per CU), so every non-MFMA phase has three other waves' MFMA streams to hide under, tiles are half as long (launch
// tail / quantisation) and a 128-ray batch -- one GPU's share of the north star's 1024-ray batch on 8 GPUs: 128 + 384
// 64-row tiles for 512 workgroup slots -- becomes 256 + 768 half tiles for 1024 slots.  Cost: every B operand float feeds
// ONE MFMA instead of two, i.e. the packed weights stream from L2 at twice the rate (16 B/clk per CU, ~9.8 TB/s aggregate),
// and the A operand is read with ds_read_b32 (one row block) instead of ds_read_b64.
//
// Tile-row mapping: MFMA row i = half-tile row i (the 64-row kernel interleaves two row blocks: row 2i + rb).  Accumulator
// registers 4t..4t+3 of lane (j, h) are rows 8t + 4h .. +3 of column j: granule g' = 2t + h of the half tile = granule
// 8T + 2t + h of the 64-row tile (T = which half).  In the 64-row fragment order that is float4 slot q = 4T + 2(t>>1) + h,
// lane' = j + 32 (t & 1) (chain_common.h frag_index), so a wave's store instruction writes two 512-byte runs.
#include <stdio.h>
#include <stdlib.h>

#include "chain_common.h"
#include "philox.h"

namespace nrf {

constexpr int HT_ROWS = 32;                       // rows per half tile
constexpr int ACT32_FLOATS = TRUNK_W * HT_ROWS;   // LDS activation tile [256][32]
constexpr int SCR32_ROWS = 12;                    // scratch rows the VALU heads need behind the activation tile (4 x 3 logit partials)

// LDS address (floats) of granule (k, g): rows 4g..4g+3 of feature k, g = 0..7.  Swizzle with (k >> 1) & 7: the epilogue's
// ds_write_b128 (16 lanes = 16 consecutive features, one granule: 128 (k & 1) + 16 (g ^ ((k >> 1) & 7)) bytes mod 256) and
// the A operand's ds_read_b32 (32 rows of an even + 32 rows of an odd feature = 256 distinct bytes) are conflict free, and
// (k >> 1) & 7 is the k-step index inside a 16-k quad, so the per-lane read offsets are the same for every quad.
__device__ __forceinline__ int act32_addr(int k, int g) { return k * HT_ROWS + 4 * (g ^ ((k >> 1) & 7)); }
__device__ __forceinline__ int act32_elem(int k, int p) { return act32_addr(k, p >> 2) + (p & 3); }

__device__ __forceinline__ float sigma_activation32(float x, int kind) {
  if (kind == 1) return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x)));   // jax.nn.softplus = logaddexp(x, 0)
  return relu(x);
}

// ---------------------------------------------------------------------------------------------
// K loops: acc[cb] (32 rows x 32 columns each) += A[32 x K] * B[K x 64] for this wave.  Weight streams as in
// chain_common.h (NCB = 2: float4 = {ks0 cb0, ks0 cb1, ks1 cb0, ks1 cb1}, 4 k per float4; quad = 16 k = 4 float4).
// ---------------------------------------------------------------------------------------------
template <int S0>
__device__ __forceinline__ void mfma32_half(f32x16 (&acc)[2], const float (&a)[4], const WQuad<2>& w) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int ks = S0 + s;
    const float4 b = w.b[ks >> 1];
    acc[0] = mfma32(a[s], (ks & 1) ? b.z : b.x, acc[0]);
    acc[1] = mfma32(a[s], (ks & 1) ? b.w : b.y, acc[1]);
  }
}

template <bool SWZ>
__device__ __forceinline__ void k_loop32(f32x16 (&acc)[2], const float* lds_in, int nquads, const float4* __restrict__ wp, int lane,
                                         const WQuad<2>& first) {
  constexpr int QUAD_FLOATS = 16 * HT_ROWS;
  asm volatile("" : "+v"(lane));   // the offsets below are recomputed per call, not kept live across the layer loop
  const int i = lane & 31, kk = lane >> 5;
  int off[8];   // per-lane float offsets of the quad's 8 A reads (k = 2t + kk)
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int k = 2 * t + kk;
    off[t] = SWZ ? act32_elem(k, i) : (k * HT_ROWS + i);
  }
  const float* ap = lds_in;
  const float4* bp = wp + lane;
  WQuad<2> bc = first;
  float a0[4], a1[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) a0[s] = ap[off[s]];
  auto quad = [&]() {
    WQuad<2> bn;   // weights run up to one quad past the end of the layer (the pack buffer is padded)
#pragma unroll
    for (int t = 0; t < 4; ++t) bn.b[t] = bp[(4 + t) * 64];
#pragma unroll
    for (int s = 0; s < 4; ++s) a1[s] = ap[off[4 + s]];
    mfma32_half<0>(acc, a0, bc);
#pragma unroll
    for (int s = 0; s < 4; ++s) a0[s] = ap[QUAD_FLOATS + off[s]];
    mfma32_half<4>(acc, a1, bc);
    __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);   // VMEM read: next quad's weights
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);   // DS read: second half of this quad
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);   // MFMA
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read (next quad)
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
    bc = bn;
    ap += QUAD_FLOATS;
    bp += 4 * 64;
  };
  if (nquads > 0) quad();   // peeled: an exact vmcnt behind the previous epilogue's stash stores (chain_common.h mfma_k_loop)
#pragma unroll 2
  for (int q = 1; q < nquads; ++q) quad();
}

// 32 columns per wave (NCB = 1 stream: float4 = ks0..ks3, 8 k per float4; quad = 2 float4).  ONE accumulator, k-steps in order:
// the same fmaf chain per output element as the 64-row kernel, so the two tilings agree bit for bit (a sub-batch of rays, which
// may run on the other tiling, reproduces its rows exactly: tests/test_gpu_fullsize.py).  The chain of dependent MFMAs costs a
// lone wave some issue slots; this layer is 5 % of a tile and the other waves of the SIMD fill them.
template <bool SWZ>
__device__ __forceinline__ void k_loop32_n1(f32x16& acc, const float* lds_in, int nquads, const float4* __restrict__ wp, int lane,
                                            const WQuad<1>& first) {
  constexpr int QUAD_FLOATS = 16 * HT_ROWS;
  asm volatile("" : "+v"(lane));
  const int i = lane & 31, kk = lane >> 5;
  int off[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int k = 2 * t + kk;
    off[t] = SWZ ? act32_elem(k, i) : (k * HT_ROWS + i);
  }
  const float* ap = lds_in;
  const float4* bp = wp + lane;
  WQuad<1> bc = first;
#pragma unroll 2
  for (int q = 0; q < nquads; ++q) {
    WQuad<1> bn;
    bn.b[0] = bp[2 * 64];
    bn.b[1] = bp[3 * 64];
    float a[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) a[s] = ap[off[s]];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const float4 b = bc.b[ks >> 2];
      const float bv = (ks & 3) == 0 ? b.x : (ks & 3) == 1 ? b.y : (ks & 3) == 2 ? b.z : b.w;
      acc = mfma32(a[ks], bv, acc);
    }
    bc = bn;
    ap += QUAD_FLOATS;
    bp += 2 * 64;
  }
}

__device__ __forceinline__ void bias_set32(f32x16 (&acc)[2], const BiasRegs<2>& r) {
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[cb][q] = r.b[cb];
}
__device__ __forceinline__ void zero32(f32x16 (&acc)[2]) {
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[cb][q] = 0.f;
}
__device__ __forceinline__ float4 piece32(const f32x16& a, int t) { return make_float4(a[4 * t], a[4 * t + 1], a[4 * t + 2], a[4 * t + 3]); }

// byte offset of piece t of lane (j, h) inside one 8 KiB feature block of a 64-row fragment tile, half T
__device__ __forceinline__ int frag32_voff(int j, int h, int t) { return (j + 32 * (t & 1)) * 16 + h * 1024; }
__device__ __forceinline__ int frag32_slot(int T, int t) { return (4 * T + 2 * (t >> 1)) * 1024; }

// Sign nibbles of a lane's four pieces (nibble t at bits 4t of nib[cb]) -> the 64-row kernel's bit image.  There, the word of
// lane (j, ho) and column block cb holds nibble q = granule (q & 1) + 2 ho + 4 (q >> 1); piece t of lane (j, h) here is granule
// 8T + 2t + h, i.e. nibble 4T + 2 (t >> 1) + h of the word of lane (j, t & 1): half T of every word belongs to this workgroup,
// and the two lanes (j, 0) / (j, 1) swap two nibbles per column block so that each writes its own lane's 16-bit half.
template <int NCB>
__device__ __forceinline__ void bits32_store(const uint32_t (&nib)[NCB], uint32_t* words_wave, int lane, int T) {
  asm volatile("" : "+v"(lane));
  const int h = lane >> 5;
  uint32_t send = 0;
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    const uint32_t n0 = nib[cb] & 15u, n1 = (nib[cb] >> 4) & 15u, n2 = (nib[cb] >> 8) & 15u, n3 = (nib[cb] >> 12) & 15u;
    send |= (h ? (n0 | (n2 << 4)) : (n1 | (n3 << 4))) << (8 * cb);
  }
  const uint32_t recv = (uint32_t)__shfl_xor((int)send, 32);
  uint16_t* out = reinterpret_cast<uint16_t*>(words_wave);
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    const uint32_t n0 = nib[cb] & 15u, n1 = (nib[cb] >> 4) & 15u, n2 = (nib[cb] >> 8) & 15u, n3 = (nib[cb] >> 12) & 15u;
    const uint32_t rc = (recv >> (8 * cb)) & 255u;
    const uint32_t half = h ? ((rc & 15u) | (n1 << 4) | ((rc >> 4) << 8) | (n3 << 12)) : (n0 | ((rc & 15u) << 4) | (n2 << 8) | ((rc >> 4) << 12));
    out[(lane * NCB + cb) * 2 + T] = (uint16_t)half;
  }
}
// the 4-bit mask of piece t of lane (., h) from the two words (lane halves 0 / 1) of its column, half T
__device__ __forceinline__ uint32_t bits32_nibble(uint32_t w_ho0, uint32_t w_ho1, int T, int t, int h) {
  return (((t & 1) ? w_ho1 : w_ho0) >> (16 * T + 4 * (2 * (t >> 1) + h))) & 15u;
}

// posenc tile [k][32 rows] in LDS -> this half of its 64-row fragment-order stash tile (chain_common.h stash_tile_from_lds)
__device__ __forceinline__ void stash32_tile_from_lds(const float* tile_lds, int kvalid, int nblocks, float* stash_tile, int T, int wave, int lane) {
  const __amdgpu_buffer_rsrc_t r = make_rsrc(stash_tile, nblocks * 32 * TILE_ROWS * 4);
  const int j = lane & 31, kk = lane >> 5;
  for (int pp = wave; pp < nblocks * 4; pp += 4) {
    const int blk = pp >> 2, qq = pp & 3;
    const int k = blk * 32 + j, g = (qq & 1) + 2 * kk + 4 * (qq >> 1);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < kvalid) v = *reinterpret_cast<const float4*>(tile_lds + k * HT_ROWS + 4 * g);
    buf_store4(v, r, lane * 16, (blk * 8 + 4 * T + qq) * 1024);
  }
}

// Layer epilogue: ReLU (or linear), LDS tile, stash, sign bits.
template <bool RELU, bool STASH>
__device__ __forceinline__ void fwd32_epilogue(f32x16 (&acc)[2], int ncol0, float* act, __amdgpu_buffer_rsrc_t stash, int stash_soff,
                                               uint32_t* bits_wave, int lane, int T) {
  asm volatile("" : "+v"(lane));   // epilogue-local lane constants: not live across the K loops
  const int j = lane & 31, h = lane >> 5;
  __syncthreads();   // every wave has finished reading the previous activations
  uint32_t nib[2] = {0u, 0u};
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const int n = ncol0 + 32 * cb + j;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float4 v = piece32(acc[cb], t);
      if (RELU) {
        if (STASH) nib[cb] |= sign_nibble(v) << (4 * t);
        v.x = relu(v.x); v.y = relu(v.y); v.z = relu(v.z); v.w = relu(v.w);
      }
      *reinterpret_cast<float4*>(act + act32_addr(n, 2 * t + h)) = v;
      if (STASH) buf_store4(v, stash, frag32_voff(j, h, t), stash_soff + cb * 8 * 1024 + frag32_slot(T, t));
    }
  }
  if (STASH && RELU) bits32_store<2>(nib, bits_wave, lane, T);
  __syncthreads();
}

struct ChainFwd32P { ChainFwdArgs a[1]; };   // read through a run-time kernarg index: scalar loads on demand (mlp_chain.hip)

template <bool STASH>
__global__ __launch_bounds__(256, 4) void nerf_mlp_fwd32_kernel(const ChainFwd32P P) {
  const ChainFwdArgs& A = P.a[blockIdx.x >> 24];
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* act = smem;                  // [256][32] swizzled
  float* pe = smem + ACT32_FLOATS;    // [max(PK, 24)][32]; scratch of the VALU heads after the skip layer
  const int tid0 = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const float* __restrict__ prm = A.params;
  const int PK = A.PK;
  const int PKS = (PK + 31) / 32 * 32;
  const int nq_pe = PK / 16;
  const int nhalf = 2 * A.ntiles;
#pragma unroll 1
  for (int ht = blockIdx.x; ht < nhalf; ht += gridDim.x) {
    int tid = tid0;
    asm volatile("" : "+v"(tid));   // per-lane constants are recomputed per tile, not hoisted (and spilled) across the kernel
    const int lane = tid & 63;
    const int j = lane & 31, h = lane >> 5;
    const int tile = ht >> 1, T = ht & 1;
    const int p = j;                     // half-tile row of the per-row (VALU) phases; 8 threads per row: part = 2 wave + h
    const int part = 2 * wave + h;
    const int row0 = tile * TILE_ROWS + HT_ROWS * T;
    // ---- prologue: sample point + SinusoidalEncoder (modules.py:213-228) ----
    {
      int r = row0 + p;
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
      auto put = [&](int k, float v) { pe[k * HT_ROWS + p] = v; };
      if (part == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) put(c, x[c]);
      } else if (part == 1) {
        for (int k = A.P; k < PK; ++k) put(k, 0.f);
      }
      const float half_pi = 1.57079632679489661923f;   // fp32(pi/2), modules.py:222
      for (int f = part; f < A.F; f += 8) {
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
    if (STASH) stash32_tile_from_lds(pe, PK, PKS / 32, A.st_pe + (size_t)tile * PKS * TILE_ROWS, T, wave, lane);

    f32x16 acc[2];
    const float4* wpk4 = reinterpret_cast<const float4*>(A.wpk);
    const size_t st_h_layer = (size_t)A.ntiles * FRAG_TILE_256;       // floats
    const int wv_soff = wave * 2 * 8 * 1024;                           // bytes: this wave's slice of a tile

    // ---- trunk: 8 x Dense(256)+ReLU, skip concat [h, posenc] at layer 4 (modules.py:41-50) ----
    const float4* wL0 = wpk4 + (A.pk.fwd_L[0] / 4) + wave * (PK / 4) * 64;
    WQuad<2> wnext = prefetch_quad<2>(wL0, lane);
    BiasRegs<2> bnext = bias_load<2>(prm + A.po.trunk_b[0], wave * 64, lane);
#pragma unroll 1
    for (int l = 0; l < TRUNK_DEPTH; ++l) {
      bias_set32(acc, bnext);
      if (l == 0) {
        k_loop32<false>(acc, pe, nq_pe, wL0, lane, wnext);
      } else {
        k_loop32<true>(acc, act, 16, wpk4 + (A.pk.fwd_L[l] / 4) + wave * 64 * 64, lane, wnext);
        if (l == A.skip) {
          const float4* w4b = wpk4 + (A.pk.fwd_L4b / 4) + wave * (PK / 4) * 64;
          k_loop32<false>(acc, pe, nq_pe, w4b, lane, prefetch_quad<2>(w4b, lane));
        }
      }
      // the next layer's first weights and bias go out before this layer's stash stores (chain_common.h bias_load)
      wnext = prefetch_quad<2>(wpk4 + ((l + 1 < TRUNK_DEPTH ? A.pk.fwd_L[l + 1] : A.pk.fwd_bn) / 4) + wave * 64 * 64, lane);
      bnext = bias_load<2>(prm + (l + 1 < TRUNK_DEPTH ? A.po.trunk_b[l + 1] : A.po.bn_b), wave * 64, lane);
      __builtin_amdgcn_sched_barrier(0);
      fwd32_epilogue<true, STASH>(
          acc, wave * 64, act,
          make_rsrc(STASH ? A.st_h + l * st_h_layer + (size_t)tile * FRAG_TILE_256 : nullptr, FRAG_TILE_256 * 4), wv_soff,
          STASH ? A.bits_trunk + (((size_t)l * A.ntiles + tile) * 4 + wave) * 128 : nullptr, lane, T);
    }

    // ---- alpha head: Dense(256->1) on the trunk output, or -- use_alpha_condition -- Dense(256+A->1) on
    //      [bottleneck, appearance code] with the per-ray code term from ray_prep (modules.py:152-157).  The SAME four partial
    //      sums as the 64-row kernel (wave w: the fmaf chain over k = 64 w .. 64 w + 63, weights wave-uniform -> s_load_dwordx16
    //      per 16 k), combined in the same order, so the two tilings give the same bits; both lane halves run the chain (32 rows
    //      fill half a wave), half 0 keeps the result ----
    float sigma_raw = 0.f;
    auto alpha_head = [&]() {
      const float4* __restrict__ wa4 = reinterpret_cast<const float4*>(prm + A.po.alpha_k) + wave * 16;
      float s = 0.f;
      const int k0 = wave * 64;
#pragma unroll 1
      for (int kc = 0; kc < 4; ++kc) {
        float4 w4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w4[i] = wa4[4 * kc + i];
        float a[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = act[act32_elem(k0 + 16 * kc + i, p)];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          s = fmaf(a[4 * i], w4[i].x, s); s = fmaf(a[4 * i + 1], w4[i].y, s);
          s = fmaf(a[4 * i + 2], w4[i].z, s); s = fmaf(a[4 * i + 3], w4[i].w, s);
        }
      }
      if (h == 0) pe[wave * HT_ROWS + p] = s;
      __syncthreads();
      if (part == 0)
        sigma_raw = (pe[p] + pe[HT_ROWS + p]) + (pe[2 * HT_ROWS + p] + pe[3 * HT_ROWS + p]) + prm[A.po.alpha_b];
    };
    if (!A.alpha_ct) alpha_head();

    // ---- bottleneck: Dense(256), no activation (modules.py:149-150) ----
    bias_set32(acc, bnext);
    k_loop32<true>(acc, act, 16, wpk4 + (A.pk.fwd_bn / 4) + wave * 64 * 64, lane, wnext);
    const float4* wrgb = wpk4 + (A.pk.fwd_rgbh / 4) + wave * 32 * 64;
    const WQuad<1> wrgb0 = prefetch_quad<1>(wrgb, lane);
    __builtin_amdgcn_sched_barrier(0);
    fwd32_epilogue<false, STASH>(acc, wave * 64, act,
                                 make_rsrc(STASH ? A.st_bn + (size_t)tile * FRAG_TILE_256 : nullptr, FRAG_TILE_256 * 4), wv_soff, nullptr,
                                 lane, T);
    if (A.alpha_ct) {
      alpha_head();   // the scratch is next written by the rgb logits, two barriers further on
      if (part == 0) sigma_raw += A.alpha_ct[min((row0 + p) / A.S, A.B - 1)];
    }

    // ---- rgb branch hidden: Dense(256+R -> 128)+ReLU; the R per-ray condition columns are folded into
    //      condterm[ray][n] (= cond . W[256:] + bias) by ray_prep ----
    {
      f32x16 acc1;
#pragma unroll
      for (int q = 0; q < 16; ++q) acc1[q] = 0.f;
      const int n = wave * 32 + j;
      // rows visited by this lane increase with t: walk the ray boundaries instead of dividing
      int ray = row0 / A.S;
      int nextb = (ray + 1) * A.S - row0;   // first half-tile row of the next ray
      float ct = A.condterm[(size_t)min(ray, A.B - 1) * RGB_W + n];
      k_loop32_n1<true>(acc1, act, 16, wrgb, lane, wrgb0);
      const __amdgpu_buffer_rsrc_t st = make_rsrc(STASH ? A.st_rgbh + (size_t)tile * FRAG_TILE_128 : nullptr, FRAG_TILE_128 * 4);
      __syncthreads();
      uint32_t nib[1] = {0u};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int g = 2 * t + h;
        const float av[4] = {acc1[4 * t], acc1[4 * t + 1], acc1[4 * t + 2], acc1[4 * t + 3]};
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int pr = 4 * g + e;
          while (pr >= nextb) { ++ray; nextb += A.S; ct = A.condterm[(size_t)min(ray, A.B - 1) * RGB_W + n]; }
          v[e] = av[e] + ct;
        }
        float4 v4 = make_float4(v[0], v[1], v[2], v[3]);
        if (STASH) nib[0] |= sign_nibble(v4) << (4 * t);
        v4.x = relu(v4.x); v4.y = relu(v4.y); v4.z = relu(v4.z); v4.w = relu(v4.w);
        *reinterpret_cast<float4*>(act + act32_addr(n, g)) = v4;
        if (STASH) buf_store4(v4, st, frag32_voff(j, h, t), wave * 8 * 1024 + frag32_slot(T, t));
      }
      if (STASH) bits32_store<1>(nib, A.bits_rgbh + ((size_t)tile * 4 + wave) * 64, lane, T);
      __syncthreads();
    }

    // ---- rgb logits Dense(128->3), sigmoid; sigma activation (models.py:276-277).  As the alpha head: the 64-row kernel's four
    //      partial sums (wave w: k = 32 w .. 32 w + 31 in order, 96 wave-uniform weights) and its order of combining them ----
    {
      const float4* __restrict__ wl4 = reinterpret_cast<const float4*>(prm + A.po.logit_k) + wave * 24;
      float sc[3] = {0.f, 0.f, 0.f};
      const int k0 = wave * 32;
#pragma unroll 1
      for (int kc = 0; kc < 2; ++kc) {   // 16 k = 48 weights per trip
        float4 w4[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) w4[i] = wl4[12 * kc + i];
        const float* wf = reinterpret_cast<const float*>(w4);
        float a[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = act[act32_elem(k0 + 16 * kc + i, p)];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          sc[0] = fmaf(a[i], wf[3 * i], sc[0]); sc[1] = fmaf(a[i], wf[3 * i + 1], sc[1]); sc[2] = fmaf(a[i], wf[3 * i + 2], sc[2]);
        }
      }
      if (h == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) pe[(3 * wave + c) * HT_ROWS + p] = sc[c];
      }
      __syncthreads();
      if (part == 0) {
        float t[3];
#pragma unroll
        for (int c = 0; c < 3; ++c)
          t[c] = (pe[c * HT_ROWS + p] + pe[(3 + c) * HT_ROWS + p]) + (pe[(6 + c) * HT_ROWS + p] + pe[(9 + c) * HT_ROWS + p]) +
                 prm[A.po.logit_b + c];
        float4 o;
        o.x = 1.f / (1.f + expf(-t[0])); o.y = 1.f / (1.f + expf(-t[1])); o.z = 1.f / (1.f + expf(-t[2]));
        if (A.noise_std > 0.f) {   // model_utils.noise_regularize (model_utils.py:266-282)
          const int row = row0 + p;
          const float nz = A.noise ? A.noise[min(row, A.rows - 1)]
                                   : philox_normal(A.dyn ? A.dyn->rng_seed : A.noise_seed, A.dyn ? A.dyn->rng_offset : A.noise_offset, A.noise_stream, (uint32_t)row);
          sigma_raw = __fadd_rn(sigma_raw, __fmul_rn(nz, A.noise_std));
        }
        o.w = sigma_activation32(sigma_raw, A.sigma_act);
        A.out4[(size_t)row0 + p] = o;
      }
      __syncthreads();   // scratch (aliases pe) is free again for the next tile's prologue
    }
  }
}

void launch_chain_fwd32(const ChainFwdArgs& a, bool stash, int grid, hipStream_t stream) {
  const size_t lds = (size_t)(ACT32_FLOATS + (a.PK > SCR32_ROWS ? a.PK : SCR32_ROWS) * HT_ROWS) * sizeof(float);
  ChainFwd32P p;
  p.a[0] = a;
  if (stash) {
    (void)hipFuncSetAttribute((const void*)nerf_mlp_fwd32_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(nerf_mlp_fwd32_kernel<true>, dim3(grid), dim3(256), lds, stream, p);
  } else {
    (void)hipFuncSetAttribute((const void*)nerf_mlp_fwd32_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(nerf_mlp_fwd32_kernel<false>, dim3(grid), dim3(256), lds, stream, p);
  }
}

// ---------------------------------------------------------------------------------------------
// backward (data gradients; bias gradients accumulated per workgroup): mlp_chain.hip bwd_tile on a half tile
// ---------------------------------------------------------------------------------------------
// small_part layout (floats): db_trunk[8][256] | db_bn[256] | db_rgbh[128] | db_logit[3] | db_alpha   (as mlp_chain.hip)
constexpr int SP32_DB_TRUNK = 0, SP32_DB_BN = 2048, SP32_DB_RGBH = 2304, SP32_DB_LOGIT = 2432, SP32_DB_ALPHA = 2435;

// Bias gradients: the 64-row kernel carries 23 per-lane partial sums across a workgroup's tiles and flushes them once per level;
// at 128 VGPRs there is no room for them next to 32 accumulators + two weight sets, so every half tile adds its column sums
// (lane halves combined by one shuffle) straight into the workgroup's OWN slice of small_part with fire-and-forget float
// atomics -- no contention (the slice is private), ~2.4 k atomics per half tile next to 330 KB of dY stores.  The host zeroes
// small_part before the launch; the reduce pass sums the slices as before.
__device__ __forceinline__ void bias32_add(float* sp, float v, int h) {
  v += __shfl_xor(v, 32);
  if (h == 0) atomicAdd(sp, v);
}

__device__ __forceinline__ void bwd32_tile(const ChainBwdArgs& A, const int ht, float* smem) {
  float* act = smem;                   // [256][32] swizzled: current dpre tile
  float* dr = smem + ACT32_FLOATS;     // [4][32]: d raw rgb (3) and d raw sigma of the half-tile rows
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = ht >> 1, T = ht & 1;
  const int row0 = tile * TILE_ROWS + HT_ROWS * T;
  const float* __restrict__ prm = A.params;
  const float4* wpk4 = reinterpret_cast<const float4*>(A.wpk);
  const size_t layer_fl = (size_t)A.ntiles * FRAG_TILE_256;   // floats per trunk layer
  const int wv = wave * 2 * 8 * 1024;                           // bytes: this wave's slice of a tile
  float* sp = A.small_part + (size_t)blockIdx.x * SMALL_PART;
  if (tid < HT_ROWS) {
    const float4 d = A.d_raw4[(size_t)row0 + tid];
    dr[tid] = d.x; dr[HT_ROWS + tid] = d.y; dr[2 * HT_ROWS + tid] = d.z; dr[3 * HT_ROWS + tid] = d.w;
  } else if (tid >= 64 && tid < 64 + HT_ROWS) {   // column sums of d raw (logit / alpha bias gradients): wave 1, lanes 0..31
    const float4 d = A.d_raw4[(size_t)row0 + tid - 64];
    float s0 = d.x, s1 = d.y, s2 = d.z, s3 = d.w;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); s3 += __shfl_xor(s3, o); }
    if (tid == 64) {
      atomicAdd(sp + SP32_DB_LOGIT, s0); atomicAdd(sp + SP32_DB_LOGIT + 1, s1); atomicAdd(sp + SP32_DB_LOGIT + 2, s2);
      atomicAdd(sp + SP32_DB_ALPHA, s3);
    }
  }
  __syncthreads();

  // ---- rgb logit^T (3 -> 128) on the VALU, ReLU mask of the rgb hidden layer ----
  {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int j = ln & 31, h = ln >> 5;
    const int n = wave * 32 + j;
    const float w0 = prm[A.po.logit_k + 3 * n], w1 = prm[A.po.logit_k + 3 * n + 1], w2 = prm[A.po.logit_k + 3 * n + 2];
    const uint32_t* bw = A.bits_rgbh + ((size_t)tile * 4 + wave) * 64;
    const uint32_t m0 = bw[j], m1 = bw[j + 32];
    const __amdgpu_buffer_rsrc_t dy = make_rsrc(A.dy_rgbh + (size_t)tile * FRAG_TILE_128, FRAG_TILE_128 * 4);
    float bsum = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int g = 2 * t + h;
      const float4 d0 = *reinterpret_cast<const float4*>(dr + 4 * g);
      const float4 d1 = *reinterpret_cast<const float4*>(dr + HT_ROWS + 4 * g);
      const float4 d2 = *reinterpret_cast<const float4*>(dr + 2 * HT_ROWS + 4 * g);
      float4 v4 = make_float4(d0.x * w0 + d1.x * w1 + d2.x * w2, d0.y * w0 + d1.y * w1 + d2.y * w2,
                              d0.z * w0 + d1.z * w1 + d2.z * w2, d0.w * w0 + d1.w * w1 + d2.w * w2);
      v4 = mask4(v4, bits32_nibble(m0, m1, T, t, h));
      bsum += (v4.x + v4.y) + (v4.z + v4.w);
      *reinterpret_cast<float4*>(act + act32_addr(n, g)) = v4;
      buf_store4(v4, dy, frag32_voff(j, h, t), wave * 8 * 1024 + frag32_slot(T, t));
    }
    bias32_add(sp + SP32_DB_RGBH + n, bsum, h);
  }
  __syncthreads();
  // ---- per-ray sums of dpre_rgbh (gradient of the per-ray condition columns of the rgb branch): thread (n, half)
  //      walks 16 half-tile rows of feature n in LDS and flushes at ray boundaries ----
  {
    int t2 = tid;
    asm volatile("" : "+v"(t2));
    const int n = t2 & 127, hf = t2 >> 7;
    const int r0 = 16 * hf;
    int ray = (row0 + r0) / A.S;
    int nextb = (ray + 1) * A.S - row0;   // first half-tile row of the next ray
    const int nvalid = A.rows - row0;      // half-tile rows >= nvalid are padding
    float ray_sum = 0.f;
#pragma unroll 1
    for (int g = r0 / 4; g < r0 / 4 + 4; ++g) {
      const float4 v4 = *reinterpret_cast<const float4*>(act + act32_addr(n, g));
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

  f32x16 acc[2];
  // ---- d bottleneck = dpre_rgbh . W_rgbh[0:256]^T   (K=128 -> N=256), linear ----
  zero32(acc);
  {
    const float4* w0 = wpk4 + (A.pk.bwd_rgbhT / 4) + wave * 32 * 64;
    k_loop32<true>(acc, act, 8, w0, lane, prefetch_quad<2>(w0, lane));
  }
  WQuad<2> wnext = prefetch_quad<2>(wpk4 + (A.pk.bwd_bnT / 4) + wave * 64 * 64, lane);
  __builtin_amdgcn_sched_barrier(0);
  {
    const __amdgpu_buffer_rsrc_t dy = make_rsrc(A.dy_bn + (size_t)tile * FRAG_TILE_256, FRAG_TILE_256 * 4);
    __syncthreads();
    int ln = lane;
    asm volatile("" : "+v"(ln));   // epilogue-local lane constants: not live across the K loops
    const int j = ln & 31, h = ln >> 5;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int n = wave * 64 + 32 * cb + j;
      const float wab = A.alpha_on_bn ? prm[A.po.alpha_k + n] : 0.f;   // use_alpha_condition: the alpha head reads the bottleneck
      float bsum = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int g = 2 * t + h;
        float4 v = piece32(acc[cb], t);
        if (A.alpha_on_bn) {
          const float4 ds = *reinterpret_cast<const float4*>(dr + 3 * HT_ROWS + 4 * g);
          v.x = fmaf(ds.x, wab, v.x); v.y = fmaf(ds.y, wab, v.y); v.z = fmaf(ds.z, wab, v.z); v.w = fmaf(ds.w, wab, v.w);
        }
        bsum += (v.x + v.y) + (v.z + v.w);
        *reinterpret_cast<float4*>(act + act32_addr(n, g)) = v;
        buf_store4(v, dy, frag32_voff(j, h, t), wv + cb * 8 * 1024 + frag32_slot(T, t));
      }
      bias32_add(sp + SP32_DB_BN + n, bsum, h);
    }
    __syncthreads();
  }

  // ---- d h8 = dbn . W_bn^T + d sigma_raw (x) w_alpha ; mask h8 > 0 -> dpre_7;  then l = 7..1 ----
#pragma unroll 1
  for (int l = TRUNK_DEPTH; l >= 1; --l) {
    zero32(acc);
    const int woff = (l == TRUNK_DEPTH) ? A.pk.bwd_bnT : A.pk.bwd_LT[l];
    k_loop32<true>(acc, act, 16, wpk4 + (woff / 4) + wave * 64 * 64, lane, wnext);
    wnext = prefetch_quad<2>(wpk4 + (A.pk.bwd_LT[l > 1 ? l - 1 : 1] / 4) + wave * 64 * 64, lane);
    __builtin_amdgcn_sched_barrier(0);
    const __amdgpu_buffer_rsrc_t dy = make_rsrc(A.dy_trunk + (size_t)(l - 1) * layer_fl + (size_t)tile * FRAG_TILE_256, FRAG_TILE_256 * 4);
    __syncthreads();
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int j = ln & 31, h = ln >> 5;
    // the output of this step is dpre_{l-1}; its mask is sign(pre_{l-1}) = bits_trunk[l-1]: the words of lanes (j, 0), (j, 1)
    const uint32_t* bw = A.bits_trunk + (((size_t)(l - 1) * A.ntiles + tile) * 4 + wave) * 128;
    const uint2 mq0 = *reinterpret_cast<const uint2*>(bw + j * 2);
    const uint2 mq1 = *reinterpret_cast<const uint2*>(bw + (j + 32) * 2);
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int n = wave * 64 + 32 * cb + j;
      const float wa = (l == TRUNK_DEPTH && !A.alpha_on_bn) ? prm[A.po.alpha_k + n] : 0.f;
      const uint32_t w0 = cb ? mq0.y : mq0.x, w1 = cb ? mq1.y : mq1.x;
      float bsum = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int g = 2 * t + h;
        float4 v = piece32(acc[cb], t);
        if (l == TRUNK_DEPTH) {
          const float4 ds = *reinterpret_cast<const float4*>(dr + 3 * HT_ROWS + 4 * g);
          v.x = fmaf(ds.x, wa, v.x); v.y = fmaf(ds.y, wa, v.y); v.z = fmaf(ds.z, wa, v.z); v.w = fmaf(ds.w, wa, v.w);
        }
        v = mask4(v, bits32_nibble(w0, w1, T, t, h));
        bsum += (v.x + v.y) + (v.z + v.w);
        *reinterpret_cast<float4*>(act + act32_addr(n, g)) = v;
        buf_store4(v, dy, frag32_voff(j, h, t), wv + cb * 8 * 1024 + frag32_slot(T, t));
      }
      bias32_add(sp + SP32_DB_TRUNK + (l - 1) * TRUNK_W + n, bsum, h);
    }
    __syncthreads();
  }
}

// ONE launch for the coarse and the fine MLP (mlp_chain.hip nerf_mlp_bwd_kernel): global half tiles [0, nt0) are level 0,
// [nt0, ntot) level 1, dealt round-robin.
struct ChainBwd32P { ChainBwdArgs a[2]; int nt0, ntot; };
__global__ __launch_bounds__(256, 4) void nerf_mlp_bwd32_kernel(const ChainBwd32P P) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int nt0 = P.nt0, ntot = P.ntot;
#pragma unroll 1
  for (int g = blockIdx.x; g < ntot; g += gridDim.x) {
    const int lv = g >= nt0 ? 1 : 0;
    bwd32_tile(P.a[lv], g - (lv ? nt0 : 0), smem);
  }
}

void launch_chain_bwd32(const ChainBwdArgs& a0, const ChainBwdArgs* a1, int grid, hipStream_t stream) {
  const size_t lds = (size_t)(ACT32_FLOATS + 4 * HT_ROWS) * sizeof(float);
  (void)hipFuncSetAttribute((const void*)nerf_mlp_bwd32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  ChainBwd32P p;
  p.a[0] = a0; p.a[1] = a1 ? *a1 : a0;
  p.nt0 = 2 * a0.ntiles; p.ntot = p.nt0 + (a1 ? 2 * a1->ntiles : 0);
  hipLaunchKernelGGL(nerf_mlp_bwd32_kernel, dim3(grid), dim3(256), lds, stream, p);
}

}  // namespace nrf
