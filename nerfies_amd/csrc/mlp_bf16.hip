// bf16-operand chains of the NeRF MLP: forward (inference / rendering, and with the training stash) and the data-gradient
// pass (BASELINE config D "bf16 MLP with fp32 composite").
//
// Opt-in mode (NRF_FLAG_BF16): activations, gradients and weights are rounded to bfloat16 (RNE) as MFMA operands; accumulation,
// biases (hi + lo bf16 pair), the per-ray condition term, the activations' ReLU and everything outside the MLP
// (sampling, compositing, loss, master weights, Adam) stay fp32.  Not bit-comparable with the fp32 path: tests bound it at
// ~1e-2 on rendered colour.
//
// Training (STASH): every layer's packed output registers -- which ARE the next layer's B operand -- are stored as they lie
// (nrf_internal.h BfStash: 1 KiB coalesced per wave store, non-temporal), plus one sign bit per pre-activation
// (v_alignbit: one VALU op per element).  The dgrad kernel below runs the same transposed chain backwards,
//   dX^T[in feature][sample] = W . dY^T,   A = W as it is stored (K = the layer's output features),
// masks with the sign bits, and stores each dpre in the same layout; wgrad_bf16.hip turns the two stashes into weight
// gradients (LDS transpose reads).
//
// Dataflow: transposed GEMMs  H^T[feature][sample] = W^T . X^T  with v_mfma_f32_32x32x16_bf16; a wave owns NG groups of 32
// samples (default 1; with 2 every weight fragment feeds two MFMAs) and ALL output features.  In the D layout lane (n, h) holds
// feature 32o + 8j + 4h + i of sample n in accumulator register 4j+i; packed to bf16 pairs these are, for k-step
// (b, s) of the next layer, exactly the B operand of lane (n, h) (k-slots 8h .. 8h+7 <-> features 32b + 8(2s+jj) + 4h + i,
// slot e = 4jj + i) -- so activations stay in registers across layers (64 packed registers per group).  Weights (A operand) are
// packed in that K order, streamed global -> LDS by LDS-DMA in chunks of 4 k-steps (double buffered, one barrier per
// chunk) and shared by the workgroup's waves.
#include <stdlib.h>

#include "chain_common.h"
#include "philox.h"

namespace nrf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;

namespace {

// BF_NOSIN / BF_NOPACK / BF_NOMFMA: compile-time switches used once to attribute the kernel's time (results are wrong with
// any of them defined): at 8192 rays x 256 samples the MFMA-free build takes 47 % of the full one, sin() 10 %, the
// bf16 pack nothing measurable.
constexpr int BF_BUF_BYTES = 5 * 9 * 1024;   // largest chunk: bias step + 4 k-steps of a 9-block GEMM

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  const f32x2 v = {a, b};
  const bf16x2 p = __builtin_convertvector(v, bf16x2);
  return __builtin_bit_cast(unsigned, p);
}

typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned relu_pk(unsigned p) {
  const s16x2 z = {0, 0};
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, p), z));
}

__device__ __forceinline__ bf16x8 as_bf16x8(unsigned a, unsigned b, unsigned c, unsigned d) {
  const u32x4v v = {a, b, c, d};
  return __builtin_bit_cast(bf16x8, v);
}

struct BfStream {   // scalars only (kept in SGPRs / VGPRs by SROA)
  const char* src;  // weight stream in global memory (this lane's byte address: base + lane * 16)
  int soff;         // byte offset of the chunk currently in LDS
  int cur;          // LDS buffer holding it
};

// Starts the LDS-DMA of `nbytes` (whole KiB) at stream offset `off` into buffer `buf`.  Every wave issues the same
// number of 1 KiB copies, ceil(KiB / waves) (the tail re-copies the last piece), so that "this wave's share of chunk c+1 has
// landed" is the compile-time test  vmcnt <= copies of chunk c+2.
__device__ __forceinline__ constexpr int bf_copies(int nbytes, int nw) { return ((nbytes >> 10) + nw - 1) / nw; }
template <int NBYTES, int NW>
__device__ __forceinline__ void bf_dma(const char* src_lane, int off, char* lds, int buf, int wave) {
  constexpr int npieces = NBYTES >> 10;
#pragma unroll
  for (int i = 0; i < bf_copies(NBYTES, NW); ++i) {
    const int p = min(wave + NW * i, npieces - 1);
    __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(src_lane + off + p * 1024),
                                     (lds_void_t*)(lds + buf * BF_BUF_BYTES + p * 1024), 16, 0, 0);
  }
}

template <int K>
__device__ __forceinline__ void bf_wait_vm() {   // s_waitcnt vmcnt(K), K a compile-time constant
  static_assert(K >= 0 && K <= 15, "vmcnt immediate");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K) : "memory");
}

// acc[g][o] = sum over the NIN input blocks (+ bias), both sample groups.  Chunk 0 carries the bias k-step first when
// BIAS.  Weights are prefetched TWO chunks ahead into a ring of three LDS buffers (an L2 -> LDS copy takes longer than one
// chunk of MFMAs): NEXT1 / NEXT2 = byte counts of the two chunks that follow this GEMM's last one in the stream; WRAP:
// they are the first two chunks of the chain (offsets 0 and NEXT1).
// bias_b0: B operand register 0 of the bias k-step (k-slots 0, 1 of the h = 0 lanes): 1, 1 for a bias; the dgrad passes
// (d sigma, d sigma) so that the row adds  d sigma * w_alpha  (the alpha head's input gradient).
template <int NG, int NW, int NIN, int NOUT, bool BIAS, int NEXT1, int NEXT2, bool WRAP = false, bool INIT = true>
__device__ __forceinline__ void bf_gemm(f32x16 (&acc)[NG][NOUT], const unsigned (&in)[NG][NIN][8], BfStream& st, char* lds, int lane,
                                        int wave, unsigned bias_b0 = 0x3F803F80u) {
  constexpr int NCHUNK = NIN / 2;   // 4 k-steps = 2 input blocks per chunk
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (INIT) {
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int o = 0; o < NOUT; ++o) acc[g][o] = zero;
  }
  // A fragments: ONE register set, refilled fragment by fragment: right after the MFMA that consumed fragment o of row r
  // has issued, the LDS read of fragment o of the NEXT row goes out into the same registers (8 MFMAs = 256 clocks ahead of its
  // use).  The chunk barrier sits in front of the LAST row of a chunk -- whose fragments are in registers by then -- so the
  // first row of the next chunk is prefetched under the last row's MFMAs as well: the LDS latency is exposed once per GEMM,
  // not once per chunk, and the fragment registers are half of a double-buffered row (the two-set version spilled ~200
  // VGPRs, whose scratch traffic also made every vmcnt wait stricter than the prefetch distance intended).
  bf16x8 af[NOUT];
  {
    const char* wb0 = lds + st.cur * BF_BUF_BYTES + lane * 16;
#pragma unroll
    for (int o = 0; o < NOUT; ++o) af[o] = *reinterpret_cast<const bf16x8*>(wb0 + o * 1024);
  }
#pragma unroll
  for (int c = 0; c < NCHUNK; ++c) {
    constexpr int BODY = 4 * NOUT * 1024;
    const int this_bytes = BODY + ((BIAS && c == 0) ? NOUT * 1024 : 0);
    // sizes / offsets of the next two chunks of the stream
    const int n1 = c + 1 < NCHUNK ? BODY : NEXT1;
    const int off1 = (WRAP && c + 1 == NCHUNK) ? 0 : st.soff + this_bytes;
    const int off2 = (WRAP && c + 2 == NCHUNK) ? 0 : (WRAP && c + 2 == NCHUNK + 1) ? NEXT1 : off1 + n1;
    const int buf1 = st.cur == 2 ? 0 : st.cur + 1;   // (cur + 1) % 3
    const int buf2 = st.cur >= 1 ? st.cur - 1 : 2;   // (cur + 2) % 3: last read in chunk c - 1, released by its barrier
    if (c + 2 < NCHUNK) bf_dma<BODY, NW>(st.src, off2, lds, buf2, wave);
    else if (c + 2 == NCHUNK) bf_dma<NEXT1, NW>(st.src, off2, lds, buf2, wave);
    else bf_dma<NEXT2, NW>(st.src, off2, lds, buf2, wave);
    const char* wb = lds + st.cur * BF_BUF_BYTES + lane * 16;
    const char* wbn = lds + buf1 * BF_BUF_BYTES + lane * 16;
    constexpr int NB = BIAS ? 1 : 0;
    const int nrows = 4 + ((BIAS && c == 0) ? 1 : 0);   // k-step rows of this chunk (bias row first)
#pragma unroll
    for (int r = 0; r < 4 + NB; ++r) {
      if (r < nrows) {
        const bool last = r + 1 == nrows;
        if (last) {
          // chunk c+1 (issued one chunk ago) must have landed; chunk c+2's copies (issued at the top of this chunk) may stay in
          // flight; this wave's own LDS reads of chunk c are complete (lgkmcnt) ...
          if (c + 2 < NCHUNK) bf_wait_vm<bf_copies(BODY, NW)>();
          else if (c + 2 == NCHUNK) bf_wait_vm<bf_copies(NEXT1, NW)>();
          else bf_wait_vm<bf_copies(NEXT2, NW)>();
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          // ... and everyone's: chunk c+1 is visible to all, and nobody reads chunk c's buffer any more (the next chunk's
          // prefetch refills it).  A bare s_barrier: __syncthreads() adds a workgroup fence, i.e. vmcnt(0), which would drain
          // the two-ahead prefetch every chunk.
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        const bool bias_row = BIAS && c == 0 && r == 0;
        const int ks = r - (nrows - 4);            // k-step inside the chunk (bias row: -1)
        const int b = 2 * c + ((ks < 0 ? 0 : ks) >> 1), s2 = (ks < 0 ? 0 : ks) & 1;
        bf16x8 bop[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g)
          bop[g] = bias_row ? as_bf16x8(bias_b0, 0u, 0u, 0u)        // B = 1 in k-slots 0, 1 (bias hi + lo)
                            : as_bf16x8(in[g][b][4 * s2], in[g][b][4 * s2 + 1], in[g][b][4 * s2 + 2], in[g][b][4 * s2 + 3]);
        const bool more = !last || c + 1 < NCHUNK;   // a next row inside this GEMM
        const char* nx = last ? wbn : wb + (r + 1) * NOUT * 1024;
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
#pragma unroll
          for (int g = 0; g < NG; ++g) {
#ifdef BF_NOMFMA
            acc[g][o][0] += __builtin_bit_cast(u32x4v, af[o]).x * 1e-30f + __builtin_bit_cast(u32x4v, bop[g]).x * 1e-30f;
#else
            acc[g][o] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[o], bop[g], acc[g][o], 0, 0, 0);
#endif
          }
          if (more) af[o] = *reinterpret_cast<const bf16x8*>(nx + o * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    st.soff = off1;
    st.cur = buf1;
  }
}

// fp32 accumulators (+ optional ReLU) -> packed bf16 B operands: register pair (4j+i, 4j+i+1) -> packed 2j + i/2
template <int NG, int NB, bool RELU>
__device__ __forceinline__ void bf_pack(unsigned (&out)[NG][NB][8], const f32x16 (&acc)[NG][NB]) {
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int o = 0; o < NB; ++o)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        float a = acc[g][o][2 * q], b = acc[g][o][2 * q + 1];
#ifdef BF_NOPACK
        out[g][o][q] = __float_as_uint(a) ^ __float_as_uint(b);
        continue;
#endif
        // ReLU AFTER the pack, on both halves at once: a negative bf16 is a negative int16 (v_pk_max_i16 with 0); rounding to
        // nearest keeps the sign, so relu(round(x)) = round(relu(x))
        const unsigned pk = pack_bf16(a, b);
        out[g][o][q] = RELU ? relu_pk(pk) : pk;
      }
}

// ---- training stash helpers ----
// NB blocks of one group: [b][jp][lane] x 16 B, non-temporal (written once, read by another kernel much later)
template <int NB>
__device__ __forceinline__ void bf_store_blocks(uint32_t* group_base, const unsigned (&v)[NB][8], int lane) {
#ifdef BF_EXP_NOSTORE   // timing attribution only (results are wrong)
  return;
#endif
  u32x4v* p = reinterpret_cast<u32x4v*>(group_base) + lane;
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      const u32x4v q = {v[b][4 * jp], v[b][4 * jp + 1], v[b][4 * jp + 2], v[b][4 * jp + 3]};
      __builtin_nontemporal_store(q, p + (b * 2 + jp) * 64);
    }
}
// ReLU derivative bits of NB blocks of accumulators: element (o, r) -> bit 31 - (16 (o & 1) + r) of dword o >> 1, 1 where
// pre > 0 (the sign bit of 0 - pre: +0 and -0 both give 0, as jax's relu gradient does); a v_sub + a v_alignbit per element
template <int NB>
__device__ __forceinline__ void bf_signbits(unsigned (&mb)[(NB + 1) / 2], const f32x16 (&acc)[NB]) {
#ifdef BF_EXP_NOBITS    // timing attribution only (results are wrong)
  return;
#endif
#pragma unroll
  for (int o = 0; o < NB; ++o)
#pragma unroll
    for (int r = 0; r < 16; ++r) mb[o >> 1] = __builtin_amdgcn_alignbit(mb[o >> 1], __float_as_uint(__fsub_rn(0.f, acc[o][r])), 31);
}
// acc = 0 where the stashed pre-activation was not positive
template <int NB>
__device__ __forceinline__ void bf_mask(f32x16 (&acc)[NB], const unsigned (&mb)[(NB + 1) / 2]) {
#ifdef BF_EXP_NOBITS
  return;
#endif
#pragma unroll
  for (int o = 0; o < NB; ++o)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int keep = __builtin_amdgcn_sbfe((int)mb[o >> 1], 31 - (16 * (o & 1) + r), 1);   // -1 where pre > 0
      acc[o][r] = __uint_as_float(__float_as_uint(acc[o][r]) & (unsigned)keep);
    }
}

__device__ __forceinline__ float bf_sigma(float x, int kind) {
  return kind == 1 ? fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))) : fmaxf(x, 0.f);
}

// chunk-0 byte counts of each GEMM of the chain (what the GEMM before it prefetches)
constexpr int KB = 1024;
constexpr int T0 = 40 * KB, T1 = 32 * KB;           // trunk GEMM: first chunk (bias row + 4 k-steps of 8 blocks), later chunks
constexpr int BN0 = 45 * KB, BN1 = 36 * KB;         // bottleneck + alpha: 9 blocks
constexpr int RG = 16 * KB;                         // rgb hidden: 4 blocks
constexpr int LG0 = 20 * KB, LG1 = 16 * KB;         // rgb logits, padded to 4 blocks so that every chunk is whole 4-KiB groups

}  // namespace

// NG sample groups of 32 per wave, NW waves per workgroup (NG * NW = 8): <2, 4> = one wave per SIMD with every weight fragment
// feeding two MFMAs; <1, 8> = two waves per SIMD (latencies overlap) at twice the LDS reads per MFMA.
template <int NG, int NW, bool STASH = false>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW / 4, NW / 4))) void nerf_mlp_fwd_bf16_kernel(const ChainFwdArgs A) {
  static_assert(!STASH || NG == 1, "the training stash is laid out for one 32-sample group per wave");
  extern __shared__ __attribute__((aligned(16))) char bf_lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 31, h = lane >> 5;
  const int niter = (A.rows + 255) / 256;   // 256 samples per workgroup iteration (4 waves x 2 groups x 32)

  BfStream st;
  st.src = reinterpret_cast<const char*>(A.wpk) + lane * 16;
  st.soff = 0;
  st.cur = 0;
  bf_dma<T0, NW>(st.src, 0, bf_lds, 0, wave);        // layer 0's only chunk
  bf_dma<T0, NW>(st.src, T0, bf_lds, 1, wave);       // layer 1, chunk 0
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

#pragma unroll 1
  for (int it = blockIdx.x; it < niter; it += gridDim.x) {
    int row[NG];
    float x[NG][3];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      row[g] = it * 256 + wave * 32 * NG + g * 32 + n;
      const int r = row[g] < A.rows ? row[g] : A.rows - 1;
      if (A.points) {
        x[g][0] = A.points[3 * r]; x[g][1] = A.points[3 * r + 1]; x[g][2] = A.points[3 * r + 2];
      } else {
        const int ray = r / A.S;
        const float z = A.zvals[r];
#pragma unroll
        for (int c = 0; c < 3; ++c) x[g][c] = __fadd_rn(A.origins[3 * ray + c], __fmul_rn(z, A.directions[3 * ray + c]));
      }
    }
    const float half_pi = 1.57079632679489661923f;
    // SinusoidalEncoder (modules.py:213-228) in fp32, packed straight into B-operand registers
    auto posenc = [&](unsigned (&pe)[NG][2][8]) {
#pragma unroll
      for (int g = 0; g < NG; ++g)
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
                val = e == 0 ? x[g][0] : e == 1 ? x[g][1] : x[g][2];
              } else if (e < A.P) {
                const int idx = e - 3, f = idx / 6, rem = idx - 6 * f, c = rem >= 3 ? rem - 3 : rem;
                const float a = __fmul_rn(c == 0 ? x[g][0] : c == 1 ? x[g][1] : x[g][2], (float)(1 << f));
#ifdef BF_NOSIN
                val = rem >= 3 ? __fadd_rn(a, half_pi) : a;
#else
                val = __sinf(rem >= 3 ? __fadd_rn(a, half_pi) : a);   // v_sin_f32: ~1e-6 abs, far below the bf16 rounding that follows
#endif
              }
              v[t] = val;
            }
            pe[g][b][q] = pack_bf16(v[0], v[1]);
          }
    };

    // training stash: this wave's group
    const size_t gidx = (size_t)it * 8 + wave;
    auto stash_layer = [&](int l, const f32x16 (&acc)[NG][8], const unsigned (&packed)[NG][8][8]) {
      if constexpr (STASH) {
        unsigned mb[4] = {0u, 0u, 0u, 0u};
        bf_signbits<8>(mb, acc[0]);
        const u32x4v q = {mb[0], mb[1], mb[2], mb[3]};
        __builtin_nontemporal_store(q, reinterpret_cast<u32x4v*>(A.bst.bits) + ((size_t)l * A.bst.ngroups + gidx) * 64 + lane);
        bf_store_blocks<8>(A.bst.h + ((size_t)l * A.bst.ngroups + gidx) * 8 * BF_BLOCK_DW, packed[0], lane);
      }
    };

    // ---- trunk ----
    unsigned act[NG][8][8];
    {
      unsigned pe[NG][2][8];
      posenc(pe);
      if constexpr (STASH) bf_store_blocks<2>(A.bst.pe + gidx * 2 * BF_BLOCK_DW, pe[0], lane);
      f32x16 acc[NG][8];
      bf_gemm<NG, NW, 2, 8, true, T0, T1>(acc, pe, st, bf_lds, lane, wave);
      bf_pack<NG, 8, true>(act, acc);
      stash_layer(0, acc, act);
    }
#pragma unroll 1
    for (int l = 1; l < TRUNK_DEPTH; ++l) {
      f32x16 acc[NG][8];
      if (l == SKIP_LAYER) {
        unsigned pe[NG][2][8];
        posenc(pe);   // before the accumulators come alive: the sin() temporaries would not fit next to 256 of them
        bf_gemm<NG, NW, 8, 8, true, T1, T0>(acc, act, st, bf_lds, lane, wave);     // then the skip rows (one 32 KiB chunk), then layer 5
        bf_gemm<NG, NW, 2, 8, false, T0, T1, false, false>(acc, pe, st, bf_lds, lane, wave);   // accumulates onto the h part
      } else if (l == TRUNK_DEPTH - 1) {
        bf_gemm<NG, NW, 8, 8, true, BN0, BN1>(acc, act, st, bf_lds, lane, wave);
      } else {
        bf_gemm<NG, NW, 8, 8, true, T0, T1>(acc, act, st, bf_lds, lane, wave);
      }
      bf_pack<NG, 8, true>(act, acc);
      stash_layer(l, acc, act);
    }

    // ---- bottleneck (linear) + alpha head as the ninth output block ----
    unsigned bn[NG][8][8];
    float alpha_raw[NG];
    {
      f32x16 acc9[NG][9];
      bf_gemm<NG, NW, 8, 9, true, RG, RG>(acc9, act, st, bf_lds, lane, wave);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        alpha_raw[g] = acc9[g][8][0];
#pragma unroll
        for (int o = 0; o < 8; ++o)
#pragma unroll
          for (int q = 0; q < 8; ++q) bn[g][o][q] = pack_bf16(acc9[g][o][2 * q], acc9[g][o][2 * q + 1]);
      }
      if constexpr (STASH) bf_store_blocks<8>(A.bst.bn + gidx * 8 * BF_BLOCK_DW, bn[0], lane);
    }

    // ---- rgb branch ----
    unsigned rgbh[NG][4][8];
    {
      f32x16 acc4[NG][4];
      bf_gemm<NG, NW, 8, 4, false, LG0, LG1>(acc4, bn, st, bf_lds, lane, wave);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int r = row[g] < A.rows ? row[g] : A.rows - 1;
        const float* ct = A.condterm + (size_t)min(r / A.S, A.B - 1) * RGB_W;   // fp32 per-ray term incl. bias
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float4 c4 = *reinterpret_cast<const float4*>(ct + 32 * o + 8 * j + 4 * h);
            acc4[g][o][4 * j] += c4.x; acc4[g][o][4 * j + 1] += c4.y; acc4[g][o][4 * j + 2] += c4.z; acc4[g][o][4 * j + 3] += c4.w;
          }
      }
      bf_pack<NG, 4, true>(rgbh, acc4);
      if constexpr (STASH) {
        unsigned mb[2] = {0u, 0u};
        bf_signbits<4>(mb, acc4[0]);
        const u32x4v q = {mb[0], mb[1], 0u, 0u};
        __builtin_nontemporal_store(q, reinterpret_cast<u32x4v*>(A.bst.bits) + ((size_t)8 * A.bst.ngroups + gidx) * 64 + lane);
        bf_store_blocks<4>(A.bst.rgbh + gidx * 4 * BF_BLOCK_DW, rgbh[0], lane);
      }
    }
    f32x16 acc1[NG][4];   // blocks 1..3 are padding (zero weights)
    bf_gemm<NG, NW, 4, 4, true, T0, T0, true>(acc1, rgbh, st, bf_lds, lane, wave);   // then the chain restarts: layer 0, layer 1
#pragma unroll
    for (int g = 0; g < NG; ++g)
      if (h == 0 && row[g] < A.rows) {
        float4 o;
        o.x = 1.f / (1.f + expf(-acc1[g][0][0]));
        o.y = 1.f / (1.f + expf(-acc1[g][0][1]));
        o.z = 1.f / (1.f + expf(-acc1[g][0][2]));
        float araw = alpha_raw[g];
        if (A.noise_std > 0.f)   // model_utils.noise_regularize (model_utils.py:266-282)
          araw += A.noise_std * (A.noise ? A.noise[row[g]]
                                         : philox_normal(A.dyn ? A.dyn->rng_seed : A.noise_seed, A.dyn ? A.dyn->rng_offset : A.noise_offset, A.noise_stream, (uint32_t)row[g]));
        o.w = bf_sigma(araw, A.sigma_act);
        A.out4[row[g]] = o;
      }
  }
}

// ---------------------------------------------------------------------------------------------
// data-gradient chain (training): d raw (rgb, sigma) -> dpre of every layer, stored for wgrad_bf16.hip
// ---------------------------------------------------------------------------------------------
namespace {
// chunk byte counts of the dgrad weight stream (GEMMs in execution order, one chunk = 4 k-steps x NOUT KiB):
//   G1  logit^T   K = 32 (3 valid) padded to one 4-k-step chunk -> 128:   DG1
//   G2  rgbh^T    K = 128 -> 256:                                          2 x DG
//   G3  bn^T      K = 256 -> 256, first chunk carries the alpha row:      DG3, 3 x DG
//   L7..L1        K = 256 -> 256:                                          4 x DG each
//   warp on, after L1:  P0  W0^T   K = 256 -> 64 (d posenc through layer 0),  P4  W4[256:]^T  K = 256 -> 64 (skip rows):  4 x DP each
constexpr int DG1 = 16 * KB, DG = 32 * KB, DG3 = 40 * KB, DP = 8 * KB;
}  // namespace

// Both levels in ONE launch (round 2: two launches of 256 and 768 iterations on 256 workgroups -- one iteration per workgroup for
// the coarse level, i.e. pure ramp): workgroups [0, n0) walk level 0, the rest level 1, split in proportion to the levels'
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
  constexpr int NW = 8;
  const int lane0 = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int niter = (A.rows + 255) / 256;
  const BfStash& S = A.st;

  BfStream st;
  st.src = reinterpret_cast<const char*>(A.wpk) + lane0 * 16;
  st.soff = 0;
  st.cur = 0;
  bf_dma<DG1, NW>(st.src, 0, bf_lds, 0, wave);      // G1
  bf_dma<DG, NW>(st.src, DG1, bf_lds, 1, wave);     // G2, chunk 0
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

#pragma unroll 1
  for (int it = wg0; it < niter; it += wgn) {
    int lo = lane0;
    asm volatile("" : "+v"(lo));   // per-iteration opaque lane: see mlp_chain.hip's backward tile
    const int lane = lo, n = lane & 31, h = lane >> 5;
    const size_t gidx = (size_t)it * 8 + wave;
    const int row = it * 256 + wave * 32 + n;
    float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < A.rows) d = A.d_raw4[row];
    auto bits_of = [&](int l) {   // nrf_internal.h BfStash::bits
      return __builtin_nontemporal_load(reinterpret_cast<const u32x4v*>(S.bits) + ((size_t)l * S.ngroups + gidx) * 64 + lane);
    };

    // ---- d raw -> the "small" dY block (features 0..3 = d rgb logits, d raw sigma) ----
    unsigned dsm[1][2][8];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < 8; ++q) dsm[0][b][q] = 0u;
    if (h == 0) { dsm[0][0][0] = pack_bf16(d.x, d.y); dsm[0][0][1] = pack_bf16(d.z, d.w); }
    bf_store_blocks<2>(S.dsmall + gidx * 2 * BF_BLOCK_DW, dsm[0], lane);
    const unsigned dsig2 = pack_bf16(d.w, d.w);

    // ---- G1: d rgb hidden = W_logit . d logits (K index 3 = d sigma meets a zero weight row), ReLU mask ----
    unsigned drg[1][4][8];
    {
      f32x16 acc4[1][4];
      bf_gemm<1, NW, 2, 4, false, DG, DG>(acc4, dsm, st, bf_lds, lane, wave);
      const u32x4v mq = bits_of(8);
      const unsigned mb[2] = {mq.x, mq.y};
      bf_mask<4>(acc4[0], mb);
      bf_pack<1, 4, false>(drg, acc4);
      bf_store_blocks<4>(S.drgbh + gidx * 4 * BF_BLOCK_DW, drg[0], lane);
    }
    // ---- G2: d bottleneck = W_rgbh[0:256] . d rgb hidden (linear) ----
    unsigned dy[1][8][8];
    {
      f32x16 acc[1][8];
      bf_gemm<1, NW, 4, 8, false, DG3, DG>(acc, drg, st, bf_lds, lane, wave);
      bf_pack<1, 8, false>(dy, acc);
      bf_store_blocks<8>(S.dbn + gidx * 8 * BF_BLOCK_DW, dy[0], lane);
    }
    // ---- G3: d h8 = W_bn . d bottleneck + w_alpha d sigma (the extra k-step), mask of layer 7 -> dpre_7 ----
    {
      f32x16 acc[1][8];
      bf_gemm<1, NW, 8, 8, true, DG, DG>(acc, dy, st, bf_lds, lane, wave, dsig2);
      const u32x4v mq = bits_of(7);
      const unsigned mb[4] = {mq.x, mq.y, mq.z, mq.w};
      bf_mask<8>(acc[0], mb);
      bf_pack<1, 8, false>(dy, acc);
      bf_store_blocks<8>(S.dy + ((size_t)7 * S.ngroups + gidx) * 8 * BF_BLOCK_DW, dy[0], lane);
    }
    // ---- l = 7..1: d h_l = W_l[0:256] . dpre_l, mask of layer l-1 -> dpre_{l-1} ----
#pragma unroll 1
    for (int l = TRUNK_DEPTH - 1; l >= 1; --l) {
      f32x16 acc[1][8];
      if (l > 1)      bf_gemm<1, NW, 8, 8, false, DG, DG>(acc, dy, st, bf_lds, lane, wave);
      else if (DPTS)  bf_gemm<1, NW, 8, 8, false, DP, DP>(acc, dy, st, bf_lds, lane, wave);          // then the d posenc GEMMs
      else            bf_gemm<1, NW, 8, 8, false, DG1, DG, true>(acc, dy, st, bf_lds, lane, wave);   // then the stream restarts: G1, G2
      const u32x4v mq = bits_of(l - 1);
      const unsigned mb[4] = {mq.x, mq.y, mq.z, mq.w};
      bf_mask<8>(acc[0], mb);
      bf_pack<1, 8, false>(dy, acc);
      bf_store_blocks<8>(S.dy + ((size_t)(l - 1) * S.ngroups + gidx) * 8 * BF_BLOCK_DW, dy[0], lane);
    }
    // ---- warp on: d posenc = W0 . dpre_0 + W4[256:] . dpre_4 (two 256 -> 64 GEMMs), chain rule through SinusoidalEncoder
    //      (modules.py:213-228) -> d points, float32, for the warp field's backward ----
    if constexpr (DPTS) {
      f32x16 ape[1][2];
      bf_gemm<1, NW, 8, 2, false, DP, DP>(ape, dy, st, bf_lds, lane, wave);   // dy = dpre_0
      {   // dpre_4 back from its stash (stored four GEMMs ago by this wave; every vmcnt wait since has retired it)
        const u32x4v* src = reinterpret_cast<const u32x4v*>(S.dy + ((size_t)SKIP_LAYER * S.ngroups + gidx) * 8 * BF_BLOCK_DW) + lane;
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
          for (int jp = 0; jp < 2; ++jp) {
            const u32x4v q = src[(b * 2 + jp) * 64];
            dy[0][b][4 * jp] = q.x; dy[0][b][4 * jp + 1] = q.y; dy[0][b][4 * jp + 2] = q.z; dy[0][b][4 * jp + 3] = q.w;
          }
      }
      bf_gemm<1, NW, 8, 2, false, DG1, DG, true, false>(ape, dy, st, bf_lds, lane, wave);   // += ; then the stream restarts
      const int r = row < A.rows ? row : A.rows - 1;
      const float x[3] = {A.points[3 * r], A.points[3 * r + 1], A.points[3 * r + 2]};
      const float half_pi = 1.57079632679489661923f;
      float dx[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
          const int e = 32 * o + 8 * (rr >> 2) + 4 * h + (rr & 3);   // posenc feature of this accumulator register
          const float g = ape[0][o][rr];
          if (e < 3) {
            dx[0] += e == 0 ? g : 0.f; dx[1] += e == 1 ? g : 0.f; dx[2] += e == 2 ? g : 0.f;
          } else if (e < A.P) {
            const int idx = e - 3, f = idx / 6, rem = idx - 6 * f, c = rem >= 3 ? rem - 3 : rem;
            const float fr = (float)(1 << f);
            const float a = __fmul_rn(c == 0 ? x[0] : c == 1 ? x[1] : x[2], fr);
            // d sin(a) = fr cos(a) = fr sin(a + pi/2);  d sin(a + pi/2) = fr sin(a + pi)
            const float dv = fr * __sinf(rem >= 3 ? __fadd_rn(a, 2.f * half_pi) : __fadd_rn(a, half_pi)) * g;
            dx[0] += c == 0 ? dv : 0.f; dx[1] += c == 1 ? dv : 0.f; dx[2] += c == 2 ? dv : 0.f;
          }
        }
#pragma unroll
      for (int c = 0; c < 3; ++c) dx[c] += __shfl_xor(dx[c], 32);   // the two lane halves hold different features of the sample
      if (h == 0 && row < A.rows_pad) {
        float* o = A.d_points + (size_t)row * 3;
        o[0] = dx[0]; o[1] = dx[1]; o[2] = dx[2];
      }
    }
  }
}

void launch_chain_bwd_bf16(const ChainBwdBf16Args& a0, const ChainBwdBf16Args* a1, int max_grid, hipStream_t stream) {
  const size_t lds = 3 * BF_BUF_BYTES;
  const int it0 = (a0.rows + 255) / 256, it1 = a1 ? (a1->rows + 255) / 256 : 0;
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
    hipLaunchKernelGGL(nerf_mlp_bwd_bf16_kernel<true>, dim3(grid), dim3(512), lds, stream, p);
  } else {
    (void)hipFuncSetAttribute((const void*)nerf_mlp_bwd_bf16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(nerf_mlp_bwd_bf16_kernel<false>, dim3(grid), dim3(512), lds, stream, p);
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
// One descriptor fills rows of the weight stream.  kind 0: `nrows` k-step rows of a GEMM (row = 2b + s), out blocks
// o0 .. o0 + nout of a GEMM that is nout_panel blocks wide; lane (m, h) gets 8 bf16: slot e <-> K index
// 32b + 8(2s + e/4) + 4h + e%4.  kind 1: the bias row: slots 0/1 of the h = 0 lanes = bf16 hi / lo parts of bias[32o + m].
__global__ __launch_bounds__(256) void bf16_pack_kernel(const RcPackDesc* __restrict__ descs, const float* __restrict__ params,
                                                        float* __restrict__ ws) {
  const RcPackDesc d = descs[blockIdx.y];
  const int total = d.ngroups * d.nout * 64;
  uint4* dst = reinterpret_cast<uint4*>(ws + d.dst_off);
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int lane = idx & 63, o = (idx >> 6) % d.nout, rowi = (idx >> 6) / d.nout;
    const int m = lane & 31, h = lane >> 5, b = rowi >> 1, s = rowi & 1;
    const int col = 32 * o + m;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (d.kind == 1) {
      if (h == 0 && col < d.ncols) {
        const float bias = params[d.src_off + col];
        const float hi = __uint_as_float(pack_bf16(bias, 0.f) << 16);
        v[0] = hi; v[1] = bias - hi;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = 32 * b + 8 * (2 * s + (e >> 2)) + 4 * h + (e & 3);
        if (k < d.krows && col < d.ncols)
          v[e] = d.transposed ? params[d.src_off + (int64_t)(d.row0 + col) * d.src_ld + k]
                              : params[d.src_off + (int64_t)(d.row0 + k) * d.src_ld + col];
      }
    }
    uint4 out;
    out.x = pack_bf16(v[0], v[1]); out.y = pack_bf16(v[2], v[3]); out.z = pack_bf16(v[4], v[5]); out.w = pack_bf16(v[6], v[7]);
    dst[(size_t)(rowi * d.nout_panel + d.o0 + o) * 64 + lane] = out;
  }
}
}  // namespace

void launch_bf16_pack(const RcPackDesc* descs, int ndesc, const float* params, float* ws, hipStream_t stream) {
  if (ndesc > 0) bf16_pack_kernel<<<dim3(32, ndesc), 256, 0, stream>>>(descs, params, ws);
}

void launch_chain_fwd_bf16(const ChainFwdArgs& a, int grid, hipStream_t stream) {
  const size_t lds = 3 * BF_BUF_BYTES;
  if (a.bst.h) {   // training: stash every layer's packed output + sign bits
    (void)hipFuncSetAttribute((const void*)nerf_mlp_fwd_bf16_kernel<1, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((nerf_mlp_fwd_bf16_kernel<1, 8, true>), dim3(grid), dim3(512), lds, stream, a);
    return;
  }
  // measured (8192 rays x 256 samples): <1, 8> 3.44 ms, <2, 4> 4.30 ms for the fine level -> two waves per SIMD by default
  static const bool w8 = getenv("NRF_BF16_W4") == nullptr;
  if (w8) {
    (void)hipFuncSetAttribute((const void*)nerf_mlp_fwd_bf16_kernel<1, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((nerf_mlp_fwd_bf16_kernel<1, 8>), dim3(grid), dim3(512), lds, stream, a);
  } else {
    (void)hipFuncSetAttribute((const void*)nerf_mlp_fwd_bf16_kernel<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((nerf_mlp_fwd_bf16_kernel<2, 4>), dim3(grid), dim3(256), lds, stream, a);
  }
}

}  // namespace nrf
