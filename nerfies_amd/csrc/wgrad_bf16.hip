// Weight-gradient GEMMs of the bf16 training path:  dW[k][n] = sum_rows X[row][k] * dY[row][n]  (and the bias gradients
// sum_rows dY[row][n]) from the two bf16 stashes the chain kernels of mlp_bf16.hip leave behind (the transpose jax.grad
// builds for modules.MLP, modules.py:41-58).
//
// Bound: HBM.  Both operands are read exactly once: 2 x 4.9 KB per MLP row = 2.6 GB per 1024 x (64+128) step, against
// 308 GFLOP (0.12 ms at the dense bf16 MFMA peak): the kernel is priced in bytes, the MFMA side only has to keep up.
//
// The stash holds, per 32-sample group and 32-feature block, the chain kernels' B operands as they lay in the registers:
// sample-major, 8 consecutive-K features per lane (nrf_internal.h BfStash).  The contraction here runs over SAMPLES, so both
// MFMA operands need 8 consecutive samples of one feature per lane -- a 16-bit transpose.  It is done by the LDS:
//   * global_load_lds copies the stash into an LDS image of 64-B chunks, chunk (n4, h, jp) = samples 4 n4 .. 4 n4 + 3 of
//     sub-block (h, jp) = 64 contiguous bytes of the stash, placed at n4 * 256 + (2 h + jp) * 64.  Four consecutive lanes of a
//     copy fetch one chunk, i.e. one 64-B request (a first version gathered single 16-B granules into a sample-major image:
//     every lane a request of its own, the L2 request rate capped the kernel at 3.0 TB/s); the LDS side is lane-linear, 1 KiB
//     per wave instruction, no registers;
//   * ds_read_b64_tr_b16 (gfx950) reads it back transposed: within a group of 16 lanes, lane c receives element c & 3 of the
//     pieces of lanes 4j + (c >> 2), j = 0..3.  With lane (r, q) of a group pointing at piece q of sample r, a lane gets
//     4 consecutive samples of feature 32 b + (lane & 31): two reads = the 8 K-slots of v_mfma_f32_32x32x16_bf16, for the A
//     operand (X) and the B operand (dY) alike.  A 32-lane half reads the four chunks of one n4: 256 bytes, every bank once.
// Work split: the stream-K tables of the fp32 kernel (nrf_internal.h WgradGroup / WgradSegment, "tile" = one 32-sample
// group); one workgroup of 8 waves holds a [Kb*32][Nb*32] fp32 partial in registers and flushes it to a slab per segment;
// reduce_kernel (wgrad.hip) sums the slabs.  Operands arrive through a 128 KiB LDS ring of 4 (8 x 8 blocks) to 10 (narrow groups)
// chunks, all but one of them in flight, one barrier per chunk.
// Round 5: an operand may be assembled from two stash buffers (WgradGroup x2_off / dy2_off), so the buffers two weight matrices
// share are streamed ONCE: the skip layer runs X = [h4 | posenc] (10 blocks) against dpre_4, the bottleneck runs h8 against
// dY = [d bottleneck | d raw] (9 blocks; column 259 of the slab is the alpha head's kernel gradient) -- rounds 2-4 read dpre_4
// and h8 twice (1.16 x the algorithmic bytes, profiles/r04_train_bf16_pmc_fetch.md).
#include "nrf_internal.h"
#include "lds_dma.h"

namespace nrf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

namespace {

constexpr int WB_LDS = 144 * 1024;           // operand ring: RING chunks (one 32-sample group each) of (Kb + Nb) x 2 KiB, X then dY;
                                             // 128 KiB for the one-source shapes, 4 x 34 / 4 x 36 KiB for the two merged ones

template <int N>
__device__ __forceinline__ void wait_vm() { wait_vmcnt<N>(); }

// operand fragment (block image at `img`, k-step ks): two transposing reads = K-slots 0..3, 4..7
__device__ __forceinline__ bf16x8 read_frag(const char* img, int ks) {
  struct { s16x4 lo, hi; } v;
  const char* p = img + ks * 1024;
  v.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  v.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 256));
  return __builtin_bit_cast(bf16x8, v);
}

// sum of the 8 bf16 of a fragment (the lane's 8 samples of one dY column)
__device__ __forceinline__ float frag_sum(const bf16x8& f) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 u = __builtin_bit_cast(u32x4, f);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += __uint_as_float(u[i] << 16) + __uint_as_float(u[i] & 0xFFFF0000u);
  return s;
}

// NRB x 2 output blocks per wave; CPW = global_load_lds instructions per wave and chunk (= ceil(2 (Kb + Nb) / 8)); RING =
// chunks the LDS ring holds (128 KiB / chunk bytes: the BYTES in flight per CU stay the same for narrow groups, whose
// chunks would otherwise be latency-bound: 2.2 us per chunk whatever its size)
template <int NRB, int NCB, int CPW, int RING>
__device__ __forceinline__ void wgrad_bf16_body(const WgradGroup& G, const WgradSegment& sg, float* ws, char* lds, int kb0, int nb0,
                                                bool active) {
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));   // per-lane constants of the seven instantiations are computed per segment, not all at kernel entry
                                  // (hoisted, 74 of them were spilled there and reloaded behind every segment's loop)
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Kb = G.Kb, Nb = G.Nb;
  const int npieces = 2 * (Kb + Nb);          // 1 KiB pieces per chunk: (operand, block, half of the samples)
  const int chunk_bytes = (Kb + Nb) * 2048;
  const char* xbase = reinterpret_cast<const char*>(ws + G.x_off);
  const char* ybase = reinterpret_cast<const char*>(ws + G.dy_off);
  const char* x2base = reinterpret_cast<const char*>(ws + G.x2_off);     // second source of X / dY blocks (== the first when unused)
  const char* y2base = reinterpret_cast<const char*>(ws + G.dy2_off);
  const int Kb1 = G.Kb1, Nb1 = G.Nb1;
  // source granule of LDS slot `lane` of a 1 KiB piece (16 samples): n = n0 + 4 (lane >> 4) + (lane & 3), h = (lane >> 3) & 1,
  // jp = (lane >> 2) & 1; stash granule (n, h, jp) of a block sits at jp * 1024 + (n + 32 h) * 16
  const unsigned src_lane = (unsigned)(((lane >> 2) & 1) * 1024 + (4 * (lane >> 4) + (lane & 3) + 32 * ((lane >> 3) & 1)) * 16);
  const unsigned lds_b = lds_byte_addr(lds);
  // The copies are asm statements (lds_dma.h): counted by hipcc, the RING - 1 chunks "in flight" were drained by a
  // compiler-inserted vmcnt(0) in front of the operand reads of every chunk.
  // Round 6: the CPW pieces a wave copies per chunk are the SAME (operand, block, half) for every chunk of a segment, so their
  // wave-uniform source addresses / strides / LDS offsets are formed once per segment and live in SGPRs; a copy is then two scalar
  // adds and the scalar-base form of the LDS-DMA (rounds 2-5 re-derived block, source buffer and a 64-bit tile offset per piece and
  // chunk: ~50 SALU instructions and six branches per piece between the chunk's barrier and its first operand read).
  const int ts_x = G.x_tile_stride, ts_x2 = G.x2_tile_stride, ts_y = G.dy_tile_stride, ts_y2 = G.dy2_tile_stride;   // values, not &G.field selects
  const char* psrc[CPW];
  int pstride[CPW];
  unsigned pdst[CPW];
#pragma unroll
  for (int i = 0; i < CPW; ++i) {
    int p = wave + 8 * i;
    p = p < npieces ? p : npieces - 1;      // the tail re-copies the last piece: every wave issues CPW copies
    const int isy = p >= 2 * Kb;
    const int pp = isy ? p - 2 * Kb : p;
    const int b = pp >> 1, half = pp & 1;
    const bool second = isy ? b >= Nb1 : b >= Kb1;
    const char* base = isy ? (second ? y2base + (b - Nb1) * 2048 : ybase + b * 2048) : (second ? x2base + (b - Kb1) * 2048 : xbase + b * 2048);
    const int stride = 4 * (isy ? (second ? ts_y2 : ts_y) : (second ? ts_x2 : ts_x));
    const char* src = base + (size_t)sg.tile_begin * stride + half * 256;   // half: samples 16..31 = 16 lanes x 16 B further
    const unsigned long long u = reinterpret_cast<unsigned long long>(src);
    psrc[i] = reinterpret_cast<const char*>(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) |
                                            (unsigned)__builtin_amdgcn_readfirstlane((int)(u & 0xffffffffu)));
    pstride[i] = __builtin_amdgcn_readfirstlane(stride);
    pdst[i] = (unsigned)__builtin_amdgcn_readfirstlane((isy ? Kb * 2048 : 0) + b * 2048 + half * 1024);
  }
  unsigned stage_buf = lds_b;               // LDS byte address of the ring slot the next staged chunk goes to
  const unsigned ring_end = lds_b + (unsigned)(RING * chunk_bytes);
  auto stage_next = [&]() __attribute__((always_inline)) {   // chunks are staged in order: the pointers walk the segment
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
      lds_dma16s<true>(psrc[i], src_lane, stage_buf + pdst[i]);
      psrc[i] += pstride[i];
    }
    stage_buf += (unsigned)chunk_bytes;
    if (stage_buf >= ring_end) stage_buf = lds_b;
  };

  f32x16 acc[NRB][NCB];
#pragma unroll
  for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][cb][r] = 0.f;
  float bsum[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) bsum[cb] = 0.f;
  const bool want_bias = G.vslab_off >= 0 && kb0 == 0 && active;

  // transposing-read base of this lane inside a block image: group (mhalf, kg) of 16 lanes, lane (r, q) in it
  const int kg = lane >> 5, mhalf = (lane >> 4) & 1, r4 = (lane >> 2) & 3, q = lane & 3;
  // piece q of sample 8 kg + r4 (+ 4 for the second read): h' = q & 1, j' = 2 mhalf + (q >> 1)  ->  chunk (n4 = 2 kg, h', jp = mhalf),
  // sample r4 of it, half jj = q >> 1
  const int frag_lane = (2 * kg) * 256 + (2 * (q & 1) + mhalf) * 64 + r4 * 16 + (q >> 1) * 8;

  const int nchunks = sg.tile_end - sg.tile_begin;
  for (int c = 0; c < RING - 1 && c < nchunks; ++c) stage_next();
  const char* rbuf = lds + frag_lane;        // this lane's read base inside the ring slot of the chunk being multiplied
  const char* const rbuf_end = lds + frag_lane + RING * chunk_bytes;
  for (int ci = 0; ci < nchunks; ++ci) {
    if (ci + RING - 2 <= nchunks - 1) wait_vm<(RING - 2) * CPW>();   // chunk ci has landed, RING - 2 later ones may fly
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();              // ... for every wave, and nobody still reads the buffer refilled next
    asm volatile("" ::: "memory");
    const bool refill = ci + RING - 1 < nchunks;
    if (active) {
      const char* buf = rbuf;
      // 10 accumulator blocks per wave (the merged skip-layer shape): the two k-steps stay a loop, so that only one k-step's operand
      // fragments are live next to the 160 accumulator registers (unrolled, hipcc hoists both steps' reads and spills 117 VGPRs)
#pragma unroll(NRB * NCB >= 10 ? 1 : 2)
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 a[NRB], b[NCB];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) a[rb] = read_frag(buf + (kb0 + rb) * 2048, ks);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) b[cb] = read_frag(buf + (Kb + nb0 + cb) * 2048, ks);
        // the refill of the slot read LAST chunk goes out behind the first operand reads: its few scalar instructions and copies issue
        // while those reads are in flight, not in front of them
        if (ks == 0 && refill) stage_next();
        if (NCB == 9) {
          // a wave = one row block x ALL column blocks: every wave reads every dY fragment, so the column sums are dealt out -- wave w
          // takes block w, wave 0 block 8 as well -- instead of piling all nine on the wave with kb0 == 0 (a chunk ends at a barrier)
          if (G.vslab_off >= 0) {
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
              if (cb == wave || (cb == 8 && wave == 0)) bsum[cb] += frag_sum(b[cb]);
          }
        } else if (want_bias) {
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) bsum[cb] += frag_sum(b[cb]);
        }
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[rb], b[cb], acc[rb][cb], 0, 0, 0);
      }
    } else if (refill) {
      stage_next();
    }
    rbuf += chunk_bytes;
    if (rbuf >= rbuf_end) rbuf = lds + frag_lane;
  }
  if (!active) return;
  const int j = lane & 31, h = lane >> 5;
  const int ld = Nb * 32;
  float* slab = ws + G.slab_off + (size_t)sg.slab_idx * (Kb * 32) * ld;
#pragma unroll
  for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int k = 32 * (kb0 + rb) + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        slab[(size_t)k * ld + 32 * (nb0 + cb) + j] = acc[rb][cb][reg];
      }
  if (NCB == 9 ? G.vslab_off >= 0 : want_bias) {   // lanes (n, kg = 0 / 1) hold different samples of column n
    float* bs = ws + G.vslab_off + (size_t)sg.slab_idx * ld;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      if (NCB == 9 && !(cb == wave || (cb == 8 && wave == 0))) continue;   // the 1 x 9 shape: the wave that summed the block
      const float t = bsum[cb] + __shfl_xor(bsum[cb], 32);
      if (h == 0) bs[32 * (nb0 + cb) + j] = t;
    }
  }
}

}  // namespace

__global__ __launch_bounds__(512) void wgrad_bf16_kernel(const WgradGroup* __restrict__ groups, const WgradSegment* __restrict__ segs,
                                                         const int* __restrict__ seg_begin, float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) char wb_lds[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int s0 = seg_begin[blockIdx.x], s1 = seg_begin[blockIdx.x + 1];
  for (int si = s0; si < s1; ++si) {
    const WgradSegment sg = segs[si];
    const WgradGroup G = groups[sg.group];
    // 8 waves tile the [Kb][Nb] block grid: n-groups of NCB column blocks, the rest along k (as wgrad.hip)
    if (G.Kb == 10 && G.Nb == 8) {          // skip layer, X = [h4 | posenc]: 4 n-groups x 2 k-groups of 5 row blocks; 36 KiB chunks
      const int wn = wave & 3, wk = wave >> 2;
      wgrad_bf16_body<5, 2, 5, 4>(G, sg, ws, wb_lds, 5 * wk, 2 * wn, true);
    } else if (G.Kb == 8 && G.Nb == 9) {    // bottleneck + alpha head, dY = [d bottleneck | d raw]: a wave = one row block x all 9; 34 KiB
      wgrad_bf16_body<1, 9, 5, 4>(G, sg, ws, wb_lds, wave, 0, true);
    } else {
    const int ngn = G.Nb / 2;            // 1, 2 or 4
    const int ngk = 8 / ngn;             // 8, 4 or 2
    const int wn = wave % ngn, wk = wave / ngn;
    const int nrb = (G.Kb + ngk - 1) / ngk;   // 4, 2 or 1
    const int kb0 = wk * nrb, nb0 = 2 * wn;
    const bool active = kb0 < G.Kb;
    const int cpw = (2 * (G.Kb + G.Nb) + 7) / 8;   // 4, 3 or 2
    if (nrb == 4)                  wgrad_bf16_body<4, 2, 4, 4>(G, sg, ws, wb_lds, kb0, nb0, active);    // 8 x 8 blocks: 32 KiB chunks
    else if (nrb == 2)             wgrad_bf16_body<2, 2, 3, 5>(G, sg, ws, wb_lds, kb0, nb0, active);    // 8 x 4: 24 KiB
    else if (cpw == 3)             wgrad_bf16_body<1, 2, 3, 6>(G, sg, ws, wb_lds, kb0, nb0, active);    // 2 x 8, 8 x 2: 20 KiB
    else if (G.Kb + G.Nb > 6)      wgrad_bf16_body<1, 2, 2, 8>(G, sg, ws, wb_lds, kb0, nb0, active);    // 4 x 4 (SE3 trunk): 16 KiB
    else                           wgrad_bf16_body<1, 2, 2, 10>(G, sg, ws, wb_lds, kb0, nb0, active);   // 4 x 2, 2 x 4: 12 KiB
    }
    __syncthreads();   // the next segment restages LDS
  }
}

void launch_wgrad_bf16(const WgradGroup* d_groups, const WgradSegment* d_segs, const int* d_seg_begin, int nwg, float* ws,
                       hipStream_t stream) {
  const size_t lds = (size_t)WB_LDS;
  (void)hipFuncSetAttribute((const void*)wgrad_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(wgrad_bf16_kernel, dim3(nwg), dim3(512), lds, stream, d_groups, d_segs, d_seg_begin, ws);
}

}  // namespace nrf
