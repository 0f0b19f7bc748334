// Device-side building blocks shared by the fused MLP chain kernels (mlp_chain.hip: 8x256 NeRF MLP;
// warp_chain.hip: 6x128 SE3 warp trunk).  gfx950 only.
//
// Tile = 64 rows (ray samples) per workgroup of 4 waves; TWO workgroups are resident per CU
// (2 x 80 KiB LDS, <= 256 VGPRs per wave), so one workgroup's layer epilogue / prologue / heads
// run under the other's MFMA stream.  A wave owns 64 rows x 64 (NCB=2) or 32 (NCB=1) columns =
// 2 MFMA row blocks x NCB column blocks of v_mfma_f32_32x32x2_f32; MFMA row block rb holds tile
// rows p = 2*i + rb (i = 0..31), so one ds_read_b64 feeds the A operand of both row blocks.
#pragma once
#include "nrf_internal.h"

namespace nrf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Stash stores go through a wave-uniform buffer descriptor (one 32-bit voffset per store instead
// of a 64-bit address).  The scalar-offset field is deliberately left at 0 and the whole offset is
// carried in the VGPR: with an SGPR soffset hipcc (ROCm 7.2) applies no "wide store data" hazard
// and schedules a VALU write of the store's data registers directly behind the
// buffer_store_dwordx4, and on gfx950 that store then picked up the NEW register contents for
// some lanes (observed: SE3 dgrad stash corrupted in exactly the component overwritten by the
// following v_pk_add_f32).  With soffset = 0 the compiler keeps the required wait state.
// The base is wave-uniform by construction, but hipcc selects 64-bit address arithmetic (layer * stride + tile * size) onto
// the VALU; the descriptor then sits in VGPRs and EVERY buffer_store is wrapped in a waterfall loop (v_readfirstlane x4,
// compare, s_and_saveexec, store, loop: 16 of them per layer epilogue of the training forward in rounds 1-2, 40 in the merged
// dgrad kernel of round 3).  Two readfirstlanes put it back into SGPRs.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, int bytes) {
  const uint64_t a = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, bytes, 0x00020000);
}
// Cache policy of the stash / dY stores (aux immediate of the buffer store: 1 = sc0, 2 = nt, 16 = sc1).  The stash is
// written once and next read by another kernel after > 1 GB of other traffic, so it is stored non-temporal: the lines
// do not displace the packed weights every workgroup re-reads from L2 (round-2 experiment, config A: 135.6 ->
// 137.9 k rays/s; with nt on the wgrad operand copies as well 138.6; write-through sc0 sc1: no change).
#ifndef NRF_STASH_AUX
#define NRF_STASH_AUX 2
#endif
__device__ __forceinline__ void buf_store4(const float4& v, __amdgpu_buffer_rsrc_t r, int voff, int off) {
  u32x4 d;
  d.x = __float_as_uint(v.x); d.y = __float_as_uint(v.y); d.z = __float_as_uint(v.z); d.w = __float_as_uint(v.w);
  __builtin_amdgcn_raw_buffer_store_b128(d, r, voff + off, 0, NRF_STASH_AUX);
}

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// LDS address (in floats) of granule (k, g): 4 consecutive tile rows 4g..4g+3 of feature k.  The
// 16 granules of a feature row are XOR-swizzled with k & 15, which makes the epilogue's
// ds_write_b128 (lanes = 32 consecutive features, one granule) bank-conflict free; the A-operand
// ds_read_b64 (lanes = the 32 row pairs of one feature) covers a whole 256-byte row either way.
__device__ __forceinline__ int act_addr(int k, int g) { return k * TILE_ROWS + 4 * (g ^ (k & 15)); }
// element (feature k, tile row p)
__device__ __forceinline__ int act_elem(int k, int p) { return act_addr(k, p >> 2) + (p & 3); }

// "Fragment" order of a [features][64 rows] tile in HBM (activation / gradient stash).  A float4
// piece = 4 consecutive rows of one feature; pieces are laid out so that (a) the epilogue of the
// chain kernels stores its accumulator registers as they lie (1 KiB coalesced per wave store) and
// (b) the wgrad kernel copies 1 KiB runs verbatim into LDS (global_load_lds) and reads them as MFMA
// operands: float4 index = ((blk*8 + q)*64 + lane), blk = feature/32, lane = feature%32 + 32*kk,
// rows 4g..4g+3 with g = (q&1) + 2*kk + 4*(q>>1).
__device__ __forceinline__ int frag_index(int k, int p) {   // float index of element (feature k, row p)
  const int g = p >> 2, q = 2 * (g >> 2) + (g & 1), kk = (g >> 1) & 1;
  return (((k >> 5) * 8 + q) * 64 + (k & 31) + 32 * kk) * 4 + (p & 3);
}

// A [features][64 rows] LDS tile (plain rows, pitch 64: the posenc / trunk-input tile of the prologues) -> its fragment-order
// stash tile in HBM, nblocks x 32 features (features >= kvalid are zero): every wave instruction stores 1 KiB contiguous.
// (Rounds 1-2 stored the prologue's stash element by element from the threads that computed it -- ~16 scattered 4-byte
// stores per thread, each behind a frag_index computation, with vmcnt(0) waits between the loops.)
__device__ __forceinline__ void stash_tile_from_lds(const float* tile_lds, int kvalid, int nblocks, float* stash_tile, int wave, int lane) {
  const __amdgpu_buffer_rsrc_t r = make_rsrc(stash_tile, nblocks * 32 * TILE_ROWS * 4);
  const int j = lane & 31, kk = lane >> 5;
  for (int pid = wave; pid < nblocks * 8; pid += 4) {
    const int blk = pid >> 3, q = pid & 7;
    const int k = blk * 32 + j, g = (q & 1) + 2 * kk + 4 * (q >> 1);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < kvalid) v = *reinterpret_cast<const float4*>(tile_lds + k * TILE_ROWS + 4 * g);
    buf_store4(v, r, lane * 16, pid * 1024);
  }
}

// ---------------------------------------------------------------------------------------------
// K loop:  acc[rb][cb] += A[64 x K] * B[K x 32*NCB]  for this wave.
//   lds_in : feature-major tile, pitch 64 floats; SWZ selects the swizzled act layout.
//   wp     : this wave's packed weights, [it][lane] float4.
//   NCB=2  : it covers 4 k  (float4 = {ks0 cb0, ks0 cb1, ks1 cb0, ks1 cb1})
//   NCB=1  : it covers 8 k  (float4 = ks0..ks3)
// The swizzle repeats every 16 k, so the loop is organised in "quads" of 16 k = 8 k-steps
// (NB = 4 / 2 weight float4s, 8 A reads, 32 / 16 MFMAs): the per-lane LDS offsets are loop
// invariant.  K must be a multiple of 16.
// ---------------------------------------------------------------------------------------------
template <int NCB> struct WQuad { float4 b[NCB == 2 ? 4 : 2]; };

template <int NCB>
__device__ __forceinline__ WQuad<NCB> prefetch_quad(const float4* __restrict__ wp, int lane) {
  WQuad<NCB> w;
#pragma unroll
  for (int q = 0; q < (NCB == 2 ? 4 : 2); ++q) w.b[q] = wp[q * 64 + lane];
  return w;
}

// MFMAs of k-steps [S0, S0+4) of a quad: a[s] = float2 (row blocks 0/1) of k-step S0+s.
template <int NCB, int S0>
__device__ __forceinline__ void mfma_half(f32x16 (&acc)[2][NCB], const float2 (&a)[4], const WQuad<NCB>& w) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int ks = S0 + s;   // k-step within the quad, 0..7
    float bv[NCB];
    if constexpr (NCB == 2) {
      const float4 b = w.b[ks >> 1];
      bv[0] = (ks & 1) ? b.z : b.x;
      bv[1] = (ks & 1) ? b.w : b.y;
    } else {
      const float4 b = w.b[ks >> 2];
      bv[0] = (ks & 3) == 0 ? b.x : (ks & 3) == 1 ? b.y : (ks & 3) == 2 ? b.z : b.w;
    }
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      acc[0][cb] = mfma32(a[s].x, bv[cb], acc[0][cb]);
      acc[1][cb] = mfma32(a[s].y, bv[cb], acc[1][cb]);
    }
  }
}

// Experiment knob (-DNRF_KLOOP_PRIO=1, scripts/build_variant.py): wave priority 0 inside the K loops, 2 everywhere else, so that a
// wave in a short VALU / LDS / barrier phase is not starved by the co-resident workgroups' MFMA streams.
#ifndef NRF_KLOOP_PRIO
#define NRF_KLOOP_PRIO 0
#endif

template <int NCB, bool SWZ>
__device__ __forceinline__ void mfma_k_loop(f32x16 (&acc)[2][NCB], const float* lds_in, int nquads,
                                            const float4* __restrict__ wp, int lane, const WQuad<NCB>& first) {
  constexpr int NB = NCB == 2 ? 4 : 2;
  constexpr int QUAD_FLOATS = 16 * TILE_ROWS;
  if (NRF_KLOOP_PRIO) __builtin_amdgcn_s_setprio(0);
  const int i = lane & 31, kk = lane >> 5;
  int off[8];   // per-lane float offsets of the quad's 8 A reads (k = 2t + kk)
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int k = 2 * t + kk;
    off[t] = SWZ ? (act_addr(k, i >> 1) + 2 * (i & 1)) : (k * TILE_ROWS + 2 * i);
  }
  const float* ap = lds_in;
  const float4* bp = wp + lane;
  WQuad<NCB> bc = first;
  float2 a0[4], a1[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) a0[s] = *reinterpret_cast<const float2*>(ap + off[s]);
  auto quad = [&]() {
    WQuad<NCB> bn;   // weights run up to one quad past the end of the layer (the pack buffer is padded)
#pragma unroll
    for (int t = 0; t < NB; ++t) bn.b[t] = bp[(NB + t) * 64];
#pragma unroll
    for (int s = 0; s < 4; ++s) a1[s] = *reinterpret_cast<const float2*>(ap + off[4 + s]);
    mfma_half<NCB, 0>(acc, a0, bc);
#pragma unroll
    for (int s = 0; s < 4; ++s) a0[s] = *reinterpret_cast<const float2*>(ap + QUAD_FLOATS + off[s]);
    mfma_half<NCB, 4>(acc, a1, bc);
    // order: weight loads, second-half A reads, first-half MFMAs with the next quad's A reads threaded in
    __builtin_amdgcn_sched_group_barrier(0x020, NB, 0);            // VMEM read
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);             // DS read x4
    __builtin_amdgcn_sched_group_barrier(0x008, 4 * NCB, 0);       // MFMA
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);           // DS read (next quad)
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * NCB, 0);     // MFMA
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 16 * NCB - 4 * NCB - 8 * NCB, 0);
    bc = bn;
    ap += QUAD_FLOATS;
    bp += NB * 64;
  };
  // The first quad is peeled out of the loop.  Inside the loop hipcc waits for the weights of the CURRENT quad with a
  // count that assumes only the loop's own loads are in flight (vmcnt(7)); entering a layer, the 16 stash stores of the
  // previous layer's epilogue are still unacknowledged, and that generic wait -- redundant in the first trip, whose weights
  // were prefetched ahead of the stores -- drained them: every layer of the training kernels exposed the store
  // acknowledgement latency.  Peeled, the first quad (32 / 16 MFMAs) runs behind an exact count and the stores retire
  // under it.
  if (nquads > 0) quad();
#pragma unroll 2
  for (int q = 1; q < nquads; ++q) quad();
  if (NRF_KLOOP_PRIO) __builtin_amdgcn_s_setprio(2);
}

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Narrow heads (N <= 32 output columns) on the MFMA pipe with K split over the workgroup's waves: this wave's slice
// C[rb][64 rows x 32 cols] = A[:, k_lo : k_lo + 32] * B, A from the swizzled LDS tile, B[k][n] = bfn(k) evaluated per lane
// (lane = column n = lane & 31, k parity lane >> 5).  16 k-steps x 2 row blocks = 32 MFMAs per wave; the caller sums the
// four waves' partials through LDS.  (VALU dot products for these heads were 12-15 % of an SE3 tile: VALU phases stretch
// 3-6x while the co-resident workgroups stream MFMAs.)
template <class BF>
__device__ __forceinline__ void mfma_kslice32(f32x16 (&acc)[2], const float* act, int k_lo, int lane, BF bfn) {
  const int i = lane & 31, kk = lane >> 5;
  float b[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) b[s] = bfn(k_lo + 2 * s + kk);
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = 0.f;
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const int k = k_lo + 2 * s + kk;
    const float2 a = *reinterpret_cast<const float2*>(act + act_addr(k, i >> 1) + 2 * (i & 1));
    acc[0] = mfma32(a.x, b[s], acc[0]);
    acc[1] = mfma32(a.y, b[s], acc[1]);
  }
}

// Tile hand-out.  counter == nullptr (default): static round-robin split.  Otherwise workgroups pull 64-row
// tiles from a global counter (zeroed by the host before the launch; `slot` is one free LDS word at the tile
// boundary).  Two workgroups share a CU and the older one wins the MFMA arbitration, so with the static split it
// finishes early and leaves the younger one alone for the last ~20 % of the kernel; the dynamic hand-out removes
// that tail but measured 4-6 % slower overall (nrf_plan.hip tile_counter_or_null), so it is off by default.
__device__ __forceinline__ int next_tile(int* __restrict__ counter, int* slot, int prev = -1) {
  if (!counter) return prev < 0 ? (int)blockIdx.x : prev + (int)gridDim.x;   // static round-robin split
  if (threadIdx.x == 0) *slot = atomicAdd(counter, 1);
  __syncthreads();
  const int t = *slot;
  __syncthreads();
  return t;
}

// Static but UNEVEN split (k_old > 0): the grid is two workgroups per CU; workgroup b < C = gridDim / 2 (dispatched first: the
// "older" one of its CU, which wins the MFMA arbitration) takes k_old of the K = ceil(ntiles / C) tiles c + k C of CU slot
// c = b mod C, the younger one the rest -- both then finish together instead of the older one leaving the younger alone
// for the last ~12 % of the kernel.  k_old = 0: the even round-robin split.
struct TileIter { int first, step, end; };
__device__ __forceinline__ TileIter tile_iter(int ntiles, int k_old) {
  if (k_old <= 0) return {(int)blockIdx.x, (int)gridDim.x, ntiles};
  const int C = gridDim.x >> 1, c = blockIdx.x % C;
  const bool old = (int)blockIdx.x < C;
  const int K = (ntiles + C - 1) / C;
  const int k0 = old ? 0 : k_old, k1 = old ? k_old : K;
  const int end = k1 * C < ntiles ? k1 * C : ntiles;
  return {c + k0 * C, C, end};
}

// acc = bias[n] broadcast down the rows: the bias add rides in the MFMA accumulator for free.
template <int NCB>
__device__ __forceinline__ void bias_acc(f32x16 (&acc)[2][NCB], const float* __restrict__ bias, int ncol0, int lane) {
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    const float bv = bias[ncol0 + 32 * cb + (lane & 31)];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][cb][r] = bv;
  }
}

// The same in two steps: the bias of the NEXT layer is fetched before this layer's epilogue issues its stash stores.  A
// load issued behind the stores can only be waited for together with them (vmcnt counts both, in order): with the bias load
// at the top of the next layer every layer of the TRAINING forward exposed the acknowledgement latency of its 16 stash
// stores, which the stash-less inference forward never saw (137 vs 123 TF for the same kernel in round 2).
template <int NCB> struct BiasRegs { float b[NCB]; };
template <int NCB>
__device__ __forceinline__ BiasRegs<NCB> bias_load(const float* __restrict__ bias, int ncol0, int lane) {
  BiasRegs<NCB> r;
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) r.b[cb] = bias[ncol0 + 32 * cb + (lane & 31)];
  return r;
}
template <int NCB>
__device__ __forceinline__ void bias_set(f32x16 (&acc)[2][NCB], const BiasRegs<NCB>& r) {
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[rb][cb][q] = r.b[cb];
}

template <int NCB>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[2][NCB]) {
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][cb][r] = 0.f;
}

// row-in-block index of accumulator register `reg` for lane half h (C/D layout of the 32x32 MFMA)
__device__ __forceinline__ int c_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }
// Accumulator registers (2q, 2q+1) x row blocks (0, 1) of one lane are 4 consecutive tile rows:
// granule g = (q&1) + 2h + 4(q>>1), q = 0..7.
__device__ __forceinline__ int q_granule(int q, int h) { return (q & 1) + 2 * h + 4 * (q >> 1); }
template <int NCB>
__device__ __forceinline__ float4 acc_piece(const f32x16 (&acc)[2][NCB], int cb, int q) {
  return make_float4(acc[0][cb][2 * q], acc[1][cb][2 * q], acc[0][cb][2 * q + 1], acc[1][cb][2 * q + 1]);
}

__device__ __forceinline__ float relu(float x) { return x > 0.f ? x : 0.f; }

// sign bits of one float4 (4 consecutive rows) -> 4-bit nibble
__device__ __forceinline__ uint32_t sign_nibble(const float4& v) {
  return (v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u);
}
__device__ __forceinline__ float4 mask4(const float4& v, uint32_t nib) {
  return make_float4((nib & 1u) ? v.x : 0.f, (nib & 2u) ? v.y : 0.f, (nib & 4u) ? v.z : 0.f, (nib & 8u) ? v.w : 0.f);
}

// Layer epilogue: write the wave's 64 x 32*NCB outputs to the LDS activation tile and, in training, to the
// fragment-order stash.  MODE 0: linear; 1: ReLU (+ 1 sign bit per element out: bits_wave[lane*NCB + cb],
// nibble q); 2: multiply by the 0/1 mask read from bits_wave (tangent pass: the ReLU derivative of the
// primal pass, warping.py:385-387 jacfwd).
enum { EPI_LINEAR = 0, EPI_RELU = 1, EPI_MASK = 2 };
template <int NCB, int MODE, bool STASH>
__device__ __forceinline__ void fwd_epilogue(f32x16 (&acc)[2][NCB], int ncol0, float* act,
                                             __amdgpu_buffer_rsrc_t stash, int stash_soff, uint32_t* bits_wave,
                                             int lane) {
  const int j = lane & 31, h = lane >> 5;
  __syncthreads();   // every wave has finished reading the previous activations
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    const int n = ncol0 + 32 * cb + j;
    uint32_t mb = MODE == EPI_MASK ? bits_wave[lane * NCB + cb] : 0u;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float4 v = acc_piece<NCB>(acc, cb, q);
      if (MODE == EPI_RELU) {
        if (STASH) mb |= sign_nibble(v) << (4 * q);
        v.x = relu(v.x); v.y = relu(v.y); v.z = relu(v.z); v.w = relu(v.w);
      } else if (MODE == EPI_MASK) {
        v = mask4(v, (mb >> (4 * q)) & 15u);
      }
      *reinterpret_cast<float4*>(act + act_addr(n, q_granule(q, h))) = v;
      if (STASH) buf_store4(v, stash, lane * 16, stash_soff + (cb * 8 + q) * 1024);
    }
    if (STASH && MODE == EPI_RELU) bits_wave[lane * NCB + cb] = mb;
  }
  __syncthreads();
}

}  // namespace nrf
