// Building blocks of the split-bf16 ("bf16x3", float32-emulating) inference chains: mlp_bf16x3.hip (NeRF MLP) and warp_bf16x3.hip (SE3
// trunk).  On top of bf16_chain.h (ring, fragment pipeline); the design notes are at the top of mlp_bf16x3.hip.  gfx950 only.
#pragma once
#include "bf16_chain.h"

namespace nrf {

constexpr int X3_SLOT = 34 * BF_KB;
constexpr int X3_LDS_BYTES = 3 * X3_SLOT;

// B operand of k-step t of a packed activation set: registers 4 (t & 1) .. + 3 of block t >> 1
template <int NB>
__device__ __forceinline__ bf16x8 kop(const unsigned (&a)[NB][8], int t) {
  return as_bf16x8(a[t >> 1][4 * (t & 1)], a[t >> 1][4 * (t & 1) + 1], a[t >> 1][4 * (t & 1) + 2], a[t >> 1][4 * (t & 1) + 3]);
}

// MFMA slots of a chunk = [BIAS: one per block,] then per k-step 3 PB: variant v = 0: W_hi . x_hi, 1: W_hi . x_lo, 2: W_lo . x_hi, each
// over the PB blocks.  Fragments of the chunk = [BIAS: PB,] then per k-step the W_hi row (PB) and the W_lo row (PB).
template <int PB, bool BIAS, int KS>
struct X3Map {
  static constexpr int NB0 = BIAS ? PB : 0;
  static constexpr int NM = NB0 + 3 * PB * KS;   // MFMAs
  static constexpr int NF = NB0 + 2 * PB * KS;   // fragments (KiB)
  static constexpr int kstep(int m) { return m < NB0 ? -1 : (m - NB0) / (3 * PB); }
  static constexpr int variant(int m) { return m < NB0 ? -1 : ((m - NB0) % (3 * PB)) / PB; }
  static constexpr int blk(int m) { return m < NB0 ? m : (m - NB0) % PB; }
  static constexpr int frag(int m) { return m < NB0 ? m : NB0 + kstep(m) * 2 * PB + (variant(m) == 2 ? PB : 0) + blk(m); }
  static constexpr bool last_use(int m) { return variant(m) != 0; }   // the bias fragment (-1) and variants 1, 2 retire their fragment
  // the MFMA slot in front of which the chunk synchronises: the first whose refill reads the NEXT chunk's slot
  static constexpr int msync() {
    for (int m = 0; m < NM; ++m)
      if (last_use(m) && frag(m) + BF_DF >= NF) return m;
    return NM;
  }
};

// One chunk of an x3 panel: acc[p] (+)= sum over the chunk's k-steps of  W_hi . x_hi + W_hi . x_lo + W_lo . x_hi  (+ the bias row).
// As bf_panel (bf16_chain.h) -- fr[] holds the chunk's first BF_DF fragments on entry and the next chunk's on exit, ONE barrier per
// chunk in front of the first refill that crosses into the next slot, the copy of chunk g+2 right behind it, epi(m) behind MFMA m --
// except that a W_hi fragment feeds two MFMAs before its register set is refilled.  bop(t, lo): B operand of k-step t.
template <int PB, bool BIAS, int KS, bool INIT, int ESPAN, int EOPS, int NW = 4, class BOp, class Epi>
__device__ __forceinline__ void x3_panel(f32x16 (&acc)[PB], ChainCtx& c, int bytes2, const bf16x8 bias_op, BOp bop, Epi epi) {
  typedef X3Map<PB, BIAS, KS> M;
  constexpr int NM = M::NM, NF = M::NF, MSYNC = M::msync();
  static_assert(NF >= BF_DF && NF * BF_KB <= X3_SLOT && ESPAN < NM, "chunk shape");
  BfRing& rg = c.rg;
  const int s1 = rg.slot == 2 ? 0 : rg.slot + 1;
  const int s2 = rg.slot == 0 ? 2 : rg.slot - 1;
  const char* cb = c.ll + rg.slot * X3_SLOT;
  const char* nb = c.ll + s1 * X3_SLOT;
  if (INIT) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < PB; ++p) acc[p] = zero;
  }
  // The refill of chunk g+2 (<= 34 pieces of 1 KiB, every wave a quarter) is NOT issued in one burst behind the barrier: a wave is alone
  // on its SIMD here, and ~9 pieces x 7 scalar / VMEM instructions in a row let the matrix pipe run dry (~250 clocks per chunk).  One
  // piece rides behind each of the MFMAs that follow the barrier; the data is needed a whole chunk later.
  const int npieces = bytes2 >> 10, full = npieces / NW, rem = npieces % NW;   // NW waves share the pieces
  const unsigned dst = rg.lds0 + (unsigned)(s2 * X3_SLOT);
#pragma unroll
  for (int m = 0; m < NM; ++m) {
    if (m == MSYNC) {
      __builtin_amdgcn_sched_barrier(0);
      bf_wait_vm<0>();                   // my pieces of chunk g+1 (copied one chunk ago) have landed
      __builtin_amdgcn_s_barrier();      // ... and everyone's; all waves are done with chunk g-1's slot
      asm volatile("" ::: "memory");
      rg.soff = __builtin_amdgcn_readfirstlane(rg.soff);
      __builtin_amdgcn_sched_barrier(0);
    }
    const int f = M::frag(m), v = M::variant(m), p = M::blk(m);
    acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c.fr[f % BF_DF], v < 0 ? bias_op : bop(M::kstep(m), v == 1), acc[p], 0, 0, 0);
    if (M::last_use(m))
      c.fr[f % BF_DF] = f + BF_DF < NF ? *reinterpret_cast<const bf16x8*>(cb + (f + BF_DF) * BF_KB)
                                       : *reinterpret_cast<const bf16x8*>(nb + (f + BF_DF - NF) * BF_KB);
    epi(m);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    if (M::last_use(m)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    if constexpr (ESPAN > 0) sgb_valu_n(epi_units_at(m, ESPAN) * EOPS + 2);
    if (m >= MSYNC) {   // piece (m - MSYNC) of this wave's share: pieces wave, wave + 4, ...
      const int j = m - MSYNC;
      static_assert(NM - MSYNC >= (X3_SLOT / BF_KB + NW - 1) / NW + 1, "the slots behind the barrier must take a wave's share of the largest chunk");
      if (j < full || (j == full && c.wave < rem)) {
        const int pc = c.wave + NW * j;
        lds_dma16s<false>(rg.src + rg.soff + pc * BF_KB, rg.voff, dst + (unsigned)(pc * BF_KB));
      }
    }
  }
  rg.soff += bytes2;
  if (rg.soff >= rg.total) rg.soff = 0;
  if constexpr (NF % BF_DF != 0) {   // slot i <- fragment i of the next chunk
    bf16x8 t[BF_DF];
#pragma unroll
    for (int i = 0; i < BF_DF; ++i) t[i] = c.fr[(i + NF) % BF_DF];
#pragma unroll
    for (int i = 0; i < BF_DF; ++i) c.fr[i] = t[i];
  }
  rg.slot = s1;
}

// VALU per epilogue unit: 2 accumulator reads (the accumulators live in AGPRs), [2 max,] pack, shift, and, packed sub, pack, and up to
// 2 writes of the results into AGPRs (a wave holds ~380 live registers: the packed sets overflow the 256 the VALU can address)
#ifndef NRF_X3_EOPS
#define NRF_X3_EOPS 9
#endif
__device__ __forceinline__ constexpr int x3_ops(bool relu) { return NRF_X3_EOPS + (relu ? 2 : 0); }
// max(x, 0) of an accumulator element.  fmaxf() canonicalises its operand first (v_max_f32 x, x: IEEE sNaN quieting; 4 VALU per pair)
// and every pure-compiler form tried (fmaxf, integer max, ReLU on the packed pair + a mask for the lo pair) ends with 45-82 spilled
// VGPRs and a 10 % slower kernel; `v_max_f32 r, 0, x` as an asm statement does not (0 spills).  But hipcc's hazard recognizer does not
// see what an asm statement reads, and nothing interlocks a VALU read behind the MFMA that writes the register: the value therefore
// passes through an identity DPP move first -- a VALU instruction the compiler knows, so the wait states behind the MFMA are its
// business -- and the asm reads the copy (+3 % against the raw asm, same-box A/B; profiles/r06_experiments.md section 5).  A whole
// unit as one volatile asm statement was tried on all three inference chains and dropped: the SE3 chain's output then depended on timing.
__device__ __forceinline__ float relu1(float x) {
  float r;
  const float y = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0xE4, 0xF, 0xF, true));
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(y));
  return r;
}

// Units of a pending panel (2 blocks = 16 register pairs) that fall on slot k: accumulators -> (ReLU) -> hi / lo bf16 pairs
template <int SPAN, int O0, bool RELU, int NBLK>
__device__ __forceinline__ void x3_epi(int k, const f32x16 (&pend)[2], unsigned (&hi)[NBLK][8], unsigned (&lo)[NBLK][8]) {
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    if (epi_slot(u, 16, SPAN) != k) continue;
    const int o = u >> 3, q = u & 7;
    float a = pend[o][2 * q], b = pend[o][2 * q + 1];
    if (RELU) { a = relu1(a); b = relu1(b); }
    const unsigned ph = pack_bf16(a, b);
    hi[O0 + o][q] = ph;
    lo[O0 + o][q] = pack_bf16(a - __uint_as_float(ph << 16), b - __uint_as_float(ph & 0xFFFF0000u));
  }
}

// PB blocks, BIAS row, KS k-steps, INIT, the riding epilogue (span, ops per unit), accumulators, size of the chunk two ahead, B operands, epilogue
#define X3_PANEL(PB, BIAS, KS, INIT, ESPAN, EOPS, ACC, B2, BOP, ...) x3_panel<PB, BIAS, KS, INIT, ESPAN, EOPS>(ACC, c, B2, bias_op, BOP, __VA_ARGS__)
// ... in a workgroup of NW waves
#define X3_PANEL_NW(NW, PB, BIAS, KS, INIT, ESPAN, EOPS, ACC, B2, BOP, ...) x3_panel<PB, BIAS, KS, INIT, ESPAN, EOPS, NW>(ACC, c, B2, bias_op, BOP, __VA_ARGS__)

}  // namespace nrf
