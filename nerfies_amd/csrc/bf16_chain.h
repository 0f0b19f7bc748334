// Building blocks of the bf16-operand NeRF MLP chains (mlp_bf16.hip): weight ring, panel GEMM, panel epilogues.  gfx950 only.
//
// Dataflow (round 4 redesign).  Transposed GEMMs  H^T[feature][sample] = W^T . X^T  with v_mfma_f32_32x32x16_bf16; a wave owns
// 32 samples and every feature, two waves per SIMD, eight per workgroup.  In the D layout lane (n, h) holds feature
// 32o + 8j + 4h + i of sample n in accumulator register 4j + i of output block o; packed to bf16 pairs these registers ARE the B
// operand of k-steps (o, 0) and (o, 1) of the next layer (k-slot e = 4jj + i of lane (n, h) <-> feature 32o + 8(2s + jj) + 4h + i),
// so activations never leave the register file.
//
// Rounds 1-3 ran a layer K-OUTER (4 k-steps of all 8 output blocks per chunk): 128 live accumulators + 64 packed inputs + 32
// fragment registers = 107-135 spilled VGPRs, and every scratch reload made hipcc drain the LDS-DMA prefetch with vmcnt(0)
// (231 of them in the inference kernel).  Now a layer runs PANEL-OUTER: a chunk of the weight stream is one PANEL of 2 output
// blocks x ALL k-steps of the layer ([row][block][lane] x 16 B; row 0 = the bias k-step), so only 2 x 16 accumulators are live,
// and the epilogue of panel p (pack, ReLU, sign bits, stash stores: ~4 VALU per packed register) is issued unit by unit
// between the MFMAs of panel p+1.  Same LDS traffic per MFMA (1 KiB), no spills, VALU work spread under the MFMA stream.
//
// Weight ring: 3 LDS slots of 42 KiB.  Chunk g+2 is copied (LDS-DMA, inline asm: invisible to hipcc's waitcnt pass) right
// behind the ONE barrier of chunk g, which sits BF_DF MFMAs before the chunk's end -- where the first fragment of chunk g+1
// is read -- so the fragment prefetch runs across chunk boundaries and the LDS latency is never exposed.  At that barrier every
// wave has issued all MFMAs of chunk g-1 (hence completed all reads of its slot: the copy target) and waited for its own pieces
// of chunk g+1 (vmcnt counts only the stash stores issued since: stores and copies retire in order).
#pragma once
#include "chain_common.h"
#include "lds_dma.h"

namespace nrf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

constexpr int BF_KB = 1024;
#ifndef NRF_BF_DF
#define NRF_BF_DF 8
#endif
constexpr int BF_DF = NRF_BF_DF;           // A fragments in flight: the LDS read of fragment f + BF_DF goes out behind MFMA f
constexpr int BF_SLOT = 42 * BF_KB;        // ring slot = the largest chunk (skip layer: 2 blocks x (1 + 16 + 4) rows)
constexpr int BF_LDS_BYTES = 3 * BF_SLOT;

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// ReLU on a packed pair: a negative bf16 is a negative int16 (v_pk_max_i16 with 0); round-to-nearest keeps the sign, so
// relu(round(x)) = round(relu(x))
__device__ __forceinline__ unsigned relu_pk(unsigned p) {
  const s16x2 z = {0, 0};
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, p), z));
}
__device__ __forceinline__ bf16x8 as_bf16x8(unsigned a, unsigned b, unsigned c, unsigned d) {
  const u32x4v v = {a, b, c, d};
  return __builtin_bit_cast(bf16x8, v);
}
// ReLU-derivative bits.  Per lane and layer 4 dwords; dword w covers the panel of blocks (2w, 2w+1); its LOW half holds the even
// accumulator registers (2q), its HIGH half the odd ones (2q + 1); unit j = 8 (o & 1) + q sits at bit 15 - j of its half.
// Forward: a post-ReLU bf16 is positive exactly where the pre-activation was (> 0), so bit = min(value as u16, 1), shifted in
// with one v_pk_mad_u16 (rounds 1-3: v_sub + v_alignbit per ELEMENT on the fp32 accumulators).
__device__ __forceinline__ unsigned bits_push(unsigned mb, unsigned relu_pair) {
  // asm: hipcc rewrites the vector form min(max(x, 0), 1) into 16-bit compares + selects + a permute on the PRE-ReLU pair (6 VALU
  // per unit, and the pre-ReLU value stays live)
  unsigned t, r;
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(t) : "v"(relu_pair), "s"(0x00010001u));
  asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(mb), "s"(0x00020002u), "v"(t));
  return r;
}
// Backward: packed pair of unit j -> the pair where the stashed bits are set, else 0
__device__ __forceinline__ unsigned bits_mask(unsigned pair, unsigned mb, int j) {
  const u16x2 sh = {(unsigned short)j, (unsigned short)j};
  const s16x2 m = __builtin_bit_cast(s16x2, (u16x2)(__builtin_bit_cast(u16x2, mb) << sh)) >> (s16x2){15, 15};
  return pair & __builtin_bit_cast(unsigned, m);
}

// (the copies use lds_dma16s, lds_dma.h: the scalar-base form of the LDS-DMA, so a piece costs two SALU adds instead of a 64-bit VALU
// add per lane, and hipcc has no per-piece address to hoist and spill)
struct BfRing {
  const char* src;   // the weight stream (wave-uniform)
  unsigned voff;     // lane * 16
  unsigned lds0;     // LDS byte address of the ring (wave-uniform)
  int soff;          // stream offset of the next chunk to copy
  int total;         // stream length (the chunk sequence is cyclic: one pass per 256-sample iteration)
  int slot;          // ring slot of the chunk being multiplied
  int turn;          // which half of the workgroup (waves 0-3 / 4-7) issues the next chunk's copies
  int nw;            // waves of the workgroup (8; fewer when a launch would leave most CUs without a workgroup: mlp_bf16.hip)
};

// `bytes` (whole KiB) of the stream at rg.soff -> ring slot `slot`, in 1-KiB pieces.
// NRF_BF_DMA_SPLIT (default): the copies of a chunk are issued by ONE HALF of the workgroup -- waves 0-3 or 4-7, i.e. one wave of
// every SIMD (a workgroup's waves go to the SIMDs cyclically) -- and the halves take turns chunk by chunk: the ~10 SALU + VMEM
// issue slots per piece sit right behind the chunk's barrier, where the two waves of a SIMD would otherwise BOTH be issuing
// copies and the matrix pipe idles; now the partner wave goes straight back to its MFMAs.  The issuing wave's vmcnt wait in
// front of the next barrier covers its pieces; the other half's wait passes at once.  0: all 8 waves take pieces round robin.
#ifndef NRF_BF_DMA_SPLIT
#define NRF_BF_DMA_SPLIT 1
#endif
template <int SLOT = BF_SLOT>
__device__ __forceinline__ void bf_ring_copy(BfRing& rg, int slot, int bytes, int wave) {
  const unsigned dst = rg.lds0 + (unsigned)(slot * SLOT);
  const int npieces = bytes >> 10;
  rg.soff = __builtin_amdgcn_readfirstlane(rg.soff);
#if NRF_BF_DMA_SPLIT
  if (rg.nw != 8) {   // a short workgroup: every wave takes pieces of every chunk
    for (int p = wave; p < npieces; p += rg.nw) lds_dma16s<false>(rg.src + rg.soff + p * BF_KB, rg.voff, dst + (unsigned)(p * BF_KB));
  } else {
    if ((wave >> 2) == rg.turn)
      for (int p = wave & 3; p < npieces; p += 4) lds_dma16s<false>(rg.src + rg.soff + p * BF_KB, rg.voff, dst + (unsigned)(p * BF_KB));
    rg.turn ^= 1;
  }
#else
  for (int p = wave; p < npieces; p += 8) lds_dma16s<false>(rg.src + rg.soff + p * BF_KB, rg.voff, dst + (unsigned)(p * BF_KB));
#endif
  rg.soff += bytes;
  if (rg.soff >= rg.total) rg.soff = 0;
}

template <int K>
__device__ __forceinline__ void bf_wait_vm() {
  static_assert(K >= 0 && K <= 63, "vmcnt immediate");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K) : "memory");
}

// One chunk = one panel: acc[p] (+)= sum over the R rows of  A[row][p] . B[row],  A fragments from the ring, B = bsel(row)
// (registers).  fr[] holds fragments 0 .. BF_DF-1 of this chunk on entry and of the next chunk on exit.  epi(k) is called behind
// MFMA k: the caller's slice of the PREVIOUS panel's epilogue; NST = VMEM stores it issues in slots < NF - BF_DF (a lower bound is
// safe, 0 always is), bytes2 = size of chunk g+2.
template <int N> __device__ __forceinline__ void sgb_valu() { if constexpr (N > 0) __builtin_amdgcn_sched_group_barrier(0x002, N, 0); }
template <int N> __device__ __forceinline__ void sgb_vmw() { if constexpr (N > 0) __builtin_amdgcn_sched_group_barrier(0x040, N, 0); }
__device__ __forceinline__ void sgb_valu_n(int n) {   // n is a constant after unrolling; the builtin wants a literal
  switch (n) {
    case 0: break;
    case 1: sgb_valu<1>(); break; case 2: sgb_valu<2>(); break; case 3: sgb_valu<3>(); break; case 4: sgb_valu<4>(); break;
    case 5: sgb_valu<5>(); break; case 6: sgb_valu<6>(); break; case 7: sgb_valu<7>(); break; case 8: sgb_valu<8>(); break;
    case 9: sgb_valu<9>(); break; case 10: sgb_valu<10>(); break; case 11: sgb_valu<11>(); break; case 12: sgb_valu<12>(); break;
    case 13: sgb_valu<13>(); break; case 14: sgb_valu<14>(); break; case 15: sgb_valu<15>(); break; case 16: sgb_valu<16>(); break;
    case 17: sgb_valu<17>(); break; case 18: sgb_valu<18>(); break; default: sgb_valu<24>(); break;
  }
}
__device__ __forceinline__ void sgb_vmw_n(int n) {
  switch (n) { case 0: break; case 1: sgb_vmw<1>(); break; case 2: sgb_vmw<2>(); break; default: sgb_vmw<4>(); break; }
}
// Spreads the NU units of a panel epilogue over the slots 1 .. SPAN of the chunk that follows the panel
__device__ __forceinline__ constexpr int epi_slot(int u, int NU, int SPAN) { return 1 + (u * SPAN) / NU; }
// schedule of a 16-unit panel epilogue: units / stash stores (one behind every 4th unit) that fall on slot k
__device__ __forceinline__ constexpr int epi_units_at(int k, int SPAN) {
  int n = 0;
  for (int u = 0; u < 16; ++u) n += epi_slot(u, 16, SPAN) == k;
  return n;
}
__device__ __forceinline__ constexpr int epi_stores_at(int k, int SPAN) {
  int n = 0;
  for (int u = 3; u < 16; u += 4) n += epi_slot(u, 16, SPAN) == k;
  return n;
}
__device__ __forceinline__ constexpr int epi_stores_before(int kend, int SPAN) {
  int n = 0;
  for (int u = 3; u < 16; u += 4) n += epi_slot(u, 16, SPAN) < kend;
  return n;
}

// ESPAN / EOPS / ESTORE: the epilogue riding in this panel (epi(k) behind MFMA k): 16 units over slots 1 .. ESPAN (0: none), VALU
// instructions per unit, whether every 4th unit is followed by a stash store.
// A chunk (= one ring slot = one barrier) may hold SEVERAL panels back to back: F0 = this panel's first fragment inside the chunk,
// NFC = the chunk's fragment count, PRE = stash stores the chunk's earlier panels issue.  The barrier + the copy of chunk g+2 sit BF_DF fragments before the CHUNK's end, in whichever
// panel that falls (bytes2 is ignored by the others); the ring slot advances behind the chunk's last panel.  Short layers (the
// 64-wide first layer, the 128-wide SE3 trunk, the heads) are merged this way: the barriers, not the MFMAs, bound them.
template <int PB, int R, bool INIT, int ESPAN, int EOPS, bool ESTORE, int F0, int NFC, int SLOT, int PRE, class BSel, class Epi>
__device__ __forceinline__ void bf_panel(f32x16 (&acc)[PB], bf16x8 (&fr)[BF_DF], BfRing& rg, const char* lds_lane, int wave,
                                         int bytes2, BSel bsel, Epi epi) {
  constexpr int NF = PB * R;
  static_assert(NFC >= BF_DF, "a chunk must hold at least BF_DF fragments");
  static_assert(F0 >= 0 && F0 + NF <= NFC, "the panel must lie inside its chunk");
  static_assert(NFC * BF_KB <= SLOT && 3 * SLOT <= 160 * BF_KB, "a chunk must fit a ring slot, three slots the LDS");
  static_assert(ESPAN < NF, "the epilogue must fit the panel");
  constexpr int KSYNC = NFC - BF_DF - F0;   // the panel-local slot in front of which the chunk synchronises (if 0 <= KSYNC < NF)
  constexpr bool LAST = F0 + NF == NFC;
  // stores this wave has issued since the copy it waits for: PRE (by the chunk's earlier panels; the caller's count, a lower bound
  // is safe) + this panel's in front of the barrier.  Counting them matters: a smaller number makes the wait cover stash stores
  // issued moments ago, i.e. an HBM write round trip (measured: the merged G1 / G2 chunk of the dgrad ran 5 % slower with PRE = 0)
  constexpr int NST = PRE + ((ESPAN > 0 && ESTORE && KSYNC >= 0 && KSYNC < NF) ? epi_stores_before(KSYNC, ESPAN) : 0);
  const int s1 = rg.slot == 2 ? 0 : rg.slot + 1;
  const int s2 = rg.slot == 0 ? 2 : rg.slot - 1;
  const char* cb = lds_lane + rg.slot * SLOT;
  const char* nb = lds_lane + s1 * SLOT;
  if (INIT) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < PB; ++p) acc[p] = zero;
  }
#pragma unroll
  for (int k = 0; k < NF; ++k) {
    if (k == KSYNC) {
      __builtin_amdgcn_sched_barrier(0);
      bf_wait_vm<NST>();                 // my pieces of chunk g+1 (copied one chunk ago) have landed; this panel's stores may fly
      __builtin_amdgcn_s_barrier();      // ... and everyone's; all waves are done with chunk g-1's slot
      asm volatile("" ::: "memory");
      bf_ring_copy<SLOT>(rg, s2, bytes2, wave);
      __builtin_amdgcn_sched_barrier(0);
    }
    const int r = k / PB, p = k % PB, f = F0 + k;
    acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[f % BF_DF], bsel(r), acc[p], 0, 0, 0);
    fr[f % BF_DF] = f + BF_DF < NFC ? *reinterpret_cast<const bf16x8*>(cb + (f + BF_DF) * BF_KB)
                                    : *reinterpret_cast<const bf16x8*>(nb + (f + BF_DF - NFC) * BF_KB);
    epi(k);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    if constexpr (ESPAN > 0) {
      sgb_valu_n(epi_units_at(k, ESPAN) * EOPS + 2);   // + slack: address arithmetic, operand moves
      if constexpr (ESTORE) sgb_vmw_n(epi_stores_at(k, ESPAN));
    }
  }
  if constexpr (LAST) {
    if constexpr (NFC % BF_DF != 0) {   // slot i <- fragment i of the next chunk
      bf16x8 t[BF_DF];
#pragma unroll
      for (int i = 0; i < BF_DF; ++i) t[i] = fr[(i + NFC) % BF_DF];
#pragma unroll
      for (int i = 0; i < BF_DF; ++i) fr[i] = t[i];
    }
    rg.slot = s1;
  }
}

// One chunk = one panel: acc[p] (+)= sum over the R rows of  A[row][p] . B[row],  A fragments from the ring, B = bsel(row)
// (registers).  fr[] holds fragments 0 .. BF_DF-1 of this chunk on entry and of the next chunk on exit.
template <int PB, int R, bool INIT, int ESPAN, int EOPS, bool ESTORE, class BSel, class Epi>
__device__ __forceinline__ void bf_chunk(f32x16 (&acc)[PB], bf16x8 (&fr)[BF_DF], BfRing& rg, const char* lds_lane, int wave,
                                         int bytes2, BSel bsel, Epi epi) {
  bf_panel<PB, R, INIT, ESPAN, EOPS, ESTORE, 0, PB * R, BF_SLOT, 0>(acc, fr, rg, lds_lane, wave, bytes2, bsel, epi);
}

// 16-byte stash store (non-temporal: written once, read by another kernel much later): buffer store with the whole offset in
// the VGPR / immediate (soffset = 0: chain_common.h hardware note)
__device__ __forceinline__ void bf_store16(__amdgpu_buffer_rsrc_t r, int voff, unsigned a, unsigned b, unsigned c, unsigned d) {
  const u32x4v q = {a, b, c, d};
  __builtin_amdgcn_raw_buffer_store_b128(q, r, voff, 0, 2);
}

// the 4 KiB of a stash group that panel `panel` (2 blocks) of a layer writes: one descriptor, immediates reach all of it
__device__ __forceinline__ __amdgpu_buffer_rsrc_t panel_rsrc(const uint32_t* group_base, int panel) {
  return make_rsrc(group_base + panel * 2 * BF_BLOCK_DW, 2 * BF_BLOCK_DW * 4);
}


struct ChainCtx {
  BfRing rg;
  bf16x8 fr[BF_DF];
  const char* ll;   // LDS ring + lane * 16
  int wave;
};


// B operand of row r of a layer whose rows are [bias,] 2 k-steps per input block: registers 4s .. 4s+3 of block b
#define BF_ROWS(arr, r0) as_bf16x8(arr[((r) - (r0)) >> 1][4 * (((r) - (r0)) & 1)], arr[((r) - (r0)) >> 1][4 * (((r) - (r0)) & 1) + 1], \
                                   arr[((r) - (r0)) >> 1][4 * (((r) - (r0)) & 1) + 2], arr[((r) - (r0)) >> 1][4 * (((r) - (r0)) & 1) + 3])


// the ring's first two chunks, the first fragments
template <int SLOT = BF_SLOT>
__device__ __forceinline__ void chain_start(ChainCtx& c, char* lds, const void* wpk, int total, int bytes0, int bytes1, int lane, int wave, int nw = 8) {
  c.rg.src = reinterpret_cast<const char*>(wpk);
  c.rg.voff = lane * 16;
  c.rg.lds0 = lds_byte_addr(lds);
  c.rg.soff = 0; c.rg.total = total; c.rg.slot = 0; c.rg.turn = 0; c.rg.nw = nw;
  c.ll = lds + lane * 16; c.wave = wave;
  bf_ring_copy<SLOT>(c.rg, 0, bytes0, wave);
  bf_ring_copy<SLOT>(c.rg, 1, bytes1, wave);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int i = 0; i < BF_DF; ++i) c.fr[i] = *reinterpret_cast<const bf16x8*>(c.ll + i * BF_KB);
}


}  // namespace nrf
