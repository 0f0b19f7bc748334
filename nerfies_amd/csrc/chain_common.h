// Device-side building blocks shared by the fused MLP chain kernels (mlp_chain.hip: 8x256 NeRF MLP;
// warp_chain.hip: 6x128 SE3 warp trunk).  gfx950 only.
#pragma once
#include "nrf_internal.h"

namespace nrf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Stash stores go through a wave-uniform buffer descriptor: the per-register offset rides in the
// scalar offset, so one voffset VGPR (lane*16) serves every store (no per-store 64-bit address).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ void buf_store4(const float4& v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
  u32x4 d;
  d.x = __float_as_uint(v.x); d.y = __float_as_uint(v.y); d.z = __float_as_uint(v.z); d.w = __float_as_uint(v.w);
  __builtin_amdgcn_raw_buffer_store_b128(d, r, voff, soff, 0);
}

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// LDS address (in floats) of granule (k, i): 4 consecutive tile rows 4i..4i+3 of feature k.
__device__ __forceinline__ int act_addr(int k, int i) { return k * TILE_ROWS + 4 * (i ^ (k & 7)); }

// acc[rb][cb] += A[128 x K] * B[K x 64(32)] for this wave.
//   lds_in : feature-major tile, pitch 128 floats; SWZ selects the swizzled act layout.
//   wp     : this wave's packed weights, [it][lane] float4.
//   NCB=2  : it covers 4 k  (float4 = {ks0 cb0, ks0 cb1, ks1 cb0, ks1 cb1}), nit = K/4
//   NCB=1  : it covers 8 k  (float4 = ks0..ks3),                               nit = K/8
template <int NCB, int KS>
__device__ __forceinline__ void mfma_block(f32x16 (&acc)[4][NCB], const float4 (&a)[KS], const float4& b) {
  const float bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const float av[4] = {a[s].x, a[s].y, a[s].z, a[s].w};
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        acc[rb][cb] = mfma32(av[rb], bv[(NCB == 2) ? (2 * s + cb) : s], acc[rb][cb]);
      }
    }
  }
}

// First weight pair of a layer.  Issued BEFORE the previous layer's epilogue so that these loads sit
// ahead of the epilogue's stash stores in the in-order vmcnt queue (gfx9 counts stores in vmcnt).
struct WPair { float4 b0, b1; };
__device__ __forceinline__ WPair prefetch_pair(const float4* __restrict__ wp, int nit, int lane) {
  WPair w;
  w.b0 = wp[lane];
  w.b1 = wp[(nit > 1 ? 1 : 0) * 64 + lane];
  return w;
}

// Software-pipelined K loop.  One "pair" = two iterations = 32 MFMAs (2048 cycles) against
// 2 weight loads (issued a full pair ahead; L2 latency under load is ~1-2k cycles) and 2*KS LDS
// A-operand reads (issued >= 16 MFMAs ahead).  The swizzle pattern repeats every 8 k, i.e. every
// pair, so the per-lane LDS offsets are loop invariant and the loop body carries no address VALU;
// sched_group_barrier spreads the loads between the MFMAs so the matrix pipe never drains.
// Weight loads run up to one pair past the end of the layer (the pack buffer is padded for it).
template <int NCB, bool SWZ>
__device__ __forceinline__ void mfma_k_loop(f32x16 (&acc)[4][NCB], const float* lds_in, int nit,
                                            const float4* __restrict__ wp, int lane, const WPair& first) {
  const int i = lane & 31, kk = lane >> 5;
  constexpr int KS = (NCB == 2) ? 2 : 4;              // k-steps (of 2 k) per iteration
  constexpr int PAIR_FLOATS = 2 * (2 * KS) * TILE_ROWS;   // LDS floats covered by one pair
  int off[2 * KS];                                     // per-lane float offsets of the pair's reads
#pragma unroll
  for (int t = 0; t < 2 * KS; ++t) {
    const int k = 2 * t + kk;
    off[t] = SWZ ? act_addr(k, i) : (k * TILE_ROWS + 4 * i);
  }
  const float* ap = lds_in;
  const float4* bp = wp + lane;
  float4 bc0 = first.b0, bc1 = first.b1;
  float4 a0[KS], a1[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) a0[s] = *reinterpret_cast<const float4*>(ap + off[s]);
  const int npairs = nit >> 1;
#pragma unroll 2
  for (int pr = 0; pr < npairs; ++pr) {
    const float4 bn0 = bp[128];
    const float4 bn1 = bp[192];
#pragma unroll
    for (int s = 0; s < KS; ++s) a1[s] = *reinterpret_cast<const float4*>(ap + off[KS + s]);
    mfma_block<NCB, KS>(acc, a0, bc0);
#pragma unroll
    for (int s = 0; s < KS; ++s) a0[s] = *reinterpret_cast<const float4*>(ap + PAIR_FLOATS + off[s]);
    mfma_block<NCB, KS>(acc, a1, bc1);
    // order: 2 weight loads, KS A reads, then MFMAs with the next-pair A reads threaded through
    __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);          // VMEM read x2
    __builtin_amdgcn_sched_group_barrier(0x100, KS, 0);         // DS read xKS (odd iteration)
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);          // MFMA x8
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);        // DS read (next even iteration)
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);        // MFMA x4
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 32 - 8 - 4 * KS, 0);
    bc0 = bn0; bc1 = bn1;
    ap += PAIR_FLOATS;
    bp += 128;
  }
  if (nit & 1) mfma_block<NCB, KS>(acc, a0, bc0);   // odd tail (K = 52: 13 iterations)
}

// acc = bias[n] broadcast down the rows: the bias add rides in the MFMA accumulator for free.
template <int NCB>
__device__ __forceinline__ void bias_acc(f32x16 (&acc)[4][NCB], const float* __restrict__ bias, int ncol0, int lane) {
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    const float bv = bias[ncol0 + 32 * cb + (lane & 31)];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][cb][r] = bv;
  }
}

template <int NCB>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[4][NCB]) {
#pragma unroll
  for (int rb = 0; rb < 4; ++rb)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][cb][r] = 0.f;
}

// row-in-block index of accumulator register `reg` for lane half h (C/D layout of 32x32 MFMA)
__device__ __forceinline__ int c_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// Epilogue-side form of act_addr(n, c_row(reg, h)): the swizzle only touches the low 3 bits of the
// granule index, so 4 per-lane offsets (one per reg&3) plus an immediate cover all 16 registers.
struct EpiAddr {
  int sw[4];
  __device__ __forceinline__ EpiAddr(int lane) {
    const int jx = lane & 7, h = lane >> 5;
#pragma unroll
    for (int q = 0; q < 4; ++q) sw[q] = 4 * ((q + 4 * h) ^ jx);
  }
  __device__ __forceinline__ int operator()(int n, int reg) const { return n * TILE_ROWS + sw[reg & 3] + 32 * (reg >> 2); }
};

__device__ __forceinline__ float relu(float x) { return x > 0.f ? x : 0.f; }

// sign bits of one float4 (4 row blocks of one accumulator register) -> 4-bit nibble
__device__ __forceinline__ uint32_t sign_nibble(const float4& v) {
  return (v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u);
}

// bits_wave: this (layer, tile, wave)'s mask words, [lane][2*NCB] dwords; dword = cb*2 + reg/8,
// nibble = reg%8, bit = row block.
template <int NCB, bool RELU, bool STASH>
__device__ __forceinline__ void fwd_epilogue(f32x16 (&acc)[4][NCB],
                                             int ncol0, float* act, __amdgpu_buffer_rsrc_t stash, int stash_soff,
                                             uint32_t* bits_wave, int lane) {
  const int j = lane & 31;
  const EpiAddr ea(lane);
  __syncthreads();   // every wave has finished reading the previous activations
  uint32_t mb[2 * NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    const int n = ncol0 + 32 * cb + j;
    mb[2 * cb] = mb[2 * cb + 1] = 0u;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      float4 v = make_float4(acc[0][cb][reg], acc[1][cb][reg], acc[2][cb][reg], acc[3][cb][reg]);
      if (RELU) {
        if (STASH) mb[2 * cb + (reg >> 3)] |= sign_nibble(v) << (4 * (reg & 7));
        v.x = relu(v.x); v.y = relu(v.y); v.z = relu(v.z); v.w = relu(v.w);
      }
      *reinterpret_cast<float4*>(act + ea(n, reg)) = v;
      if (STASH) buf_store4(v, stash, lane * 16, stash_soff + (cb * 16 + reg) * 1024);
    }
  }
  if (STASH && RELU) {
#pragma unroll
    for (int q = 0; q < 2 * NCB; ++q) bits_wave[lane * (2 * NCB) + q] = mb[q];
  }
  __syncthreads();
}

__device__ __forceinline__ float4 mask4(const float4& v, uint32_t nib) {
  return make_float4((nib & 1u) ? v.x : 0.f, (nib & 2u) ? v.y : 0.f, (nib & 4u) ? v.z : 0.f, (nib & 8u) ? v.w : 0.f);
}

}  // namespace nrf
