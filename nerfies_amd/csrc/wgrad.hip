// Weight-gradient GEMMs  dW[k][n] = sum_rows X[row][k] * dY[row][n]  for every dense layer
// of the NeRF MLPs and of the SE3 trunk (the transpose jax.grad builds for modules.MLP,
// modules.py:41-58).
//
// The reduction runs over rows (ray samples): 65k-196k per MLP, output only 256x256.  Each
// workgroup walks a stream-K share of the (layer, 64-row tile) work list and, per segment, holds
// a full [Kb*32][Nb*32] partial in registers: 8 waves x (<=4 x 2) 32x32 fp32 MFMA blocks, so X
// and dY stream from HBM exactly once.  Both operands live in HBM in "fragment" order
// (chain_common.h): a 32-row chunk of 32 features is a run of 4 KiB that is copied VERBATIM into
// LDS with global_load_lds (1 KiB per wave instruction, no staging registers, no ds_write, no
// swizzle) and read back as MFMA operands with conflict-free ds_read_b128: the float4 a lane
// receives is 4 consecutive rows = the k of 4 MFMA steps, identical for X and dY.
//
// Operand ring (round 3): a chunk (32 rows of X and dY) is (Kb + Nb) x 4 KiB -- 64 KiB for a 256 x 256 layer but only
// 32 / 24 KiB for the 128 x 128 and 64 x 128 groups of the SE3 trunk, whose 32 (16) MFMAs per wave and chunk are over in
// ~1.7 us: with two stages the copies of the next chunk were issued one chunk ahead, less than the HBM latency under load,
// and those groups ran latency-bound at a third of the wide groups' rate.  The ring now always fills the CU's 160 KiB:
// RING = 2 stages for 8 x 8 blocks up to 6 for 2 x 4, all but one in flight (s_waitcnt vmcnt(N) counts the copies of the
// later chunks), one barrier per chunk.  Partials go to slabs, summed by reduce_kernel.
#include "nrf_internal.h"
#include "lds_dma.h"

namespace nrf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WG_LDS_BYTES = 160 * 1024;      // the whole LDS of a CU: one workgroup per CU
constexpr int WG_PIECE = 256;                 // floats per 1 KiB copy piece (one wave instruction)

// cache policy of the streamed operand copies: non-temporal (every byte is read exactly once); NRF_WGRAD_AUX=0: default policy
#ifndef NRF_WGRAD_AUX
#define NRF_WGRAD_AUX 2
#endif
constexpr bool WG_NT = NRF_WGRAD_AUX == 2;

template <int N>
__device__ __forceinline__ void wait_vm() { wait_vmcnt<N>(); }

// Issues this wave's CPW copies of chunk `c` (32 rows) of one tile into the stage at `buf`: pieces p < 4 Kb are X (block
// p >> 2, quarter p & 3), the rest dY; piece p lands at buf + p * 1 KiB and is dealt round-robin to the 8 waves.  Every
// wave issues exactly CPW copies (the tail re-copies the last piece) so that one vmcnt count fits all waves.  The copies are
// asm statements (lds_dma.h): hipcc must not count them, or it drains the ring in front of the operand reads.
template <int CPW>
__device__ __forceinline__ void stage_chunk(const float* __restrict__ xt, int Kb, const float* __restrict__ yt, int Nb, int c,
                                            unsigned buf, int wave, int lane) {
  const int npx = Kb * 4, np = (Kb + Nb) * 4;
#pragma unroll
  for (int i = 0; i < CPW; ++i) {
    int p = wave + 8 * i;
    p = p < np ? p : np - 1;
    const bool isy = p >= npx;
    const int pp = isy ? p - npx : p;
    const int blk = pp >> 2, qq = pp & 3;
    const float* src = (isy ? yt : xt) + ((size_t)(blk * 8 + 4 * c + qq) * 64 + lane) * 4;
    lds_dma16<WG_NT>(src, buf + p * (WG_PIECE * 4));   // wave-uniform LDS base; the hardware adds lane * 16 bytes
  }
}

// `refill()` (the copies of the chunk RING - 1 ahead) is called behind the MFMAs of quarter `when` of the chunk: the few scalar
// instructions and copies of a wave issue while its own last MFMA and its SIMD partner's MFMAs occupy the matrix pipe.
template <int NRB, class Refill>
__device__ __forceinline__ void wgrad_compute(f32x16 (&acc)[NRB][2], const float* Xs, const float* Ys,
                                              int kb0, int nb0, int lane, int when, Refill refill) {
  float4 a[2][NRB], b[2][2];
  auto load = [&](int qq, int buf) {
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
      a[buf][rb] = *reinterpret_cast<const float4*>(Xs + (((kb0 + rb) * 4 + qq) * 64 + lane) * 4);
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
      b[buf][cb] = *reinterpret_cast<const float4*>(Ys + (((nb0 + cb) * 4 + qq) * 64 + lane) * 4);
  };
  load(0, 0);
#pragma unroll
  for (int qq = 0; qq < 4; ++qq) {
    const int cur = qq & 1;
    if (qq + 1 < 4) load(qq + 1, cur ^ 1);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) {
        const float4 av4 = a[cur][rb];
        const float av = s == 0 ? av4.x : s == 1 ? av4.y : s == 2 ? av4.z : av4.w;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          const float4 bv4 = b[cur][cb];
          const float bv = s == 0 ? bv4.x : s == 1 ? bv4.y : s == 2 ? bv4.z : bv4.w;
          acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[rb][cb], 0, 0, 0);
        }
      }
    }
    if (qq < 2 && qq == when) refill();
  }
}

// NRB x 2 output blocks per wave, CPW copies per wave and chunk (= ceil(4 (Kb + Nb) / 8)), RING stages of
// 4 (Kb + Nb) KiB (RING x stage <= 160 KiB).
template <int NRB, int CPW, int RING>
__device__ __forceinline__ void wgrad_body(const WgradTask& T, float* smem, int kb0, int nb0) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  f32x16 acc[NRB][2];
#pragma unroll
  for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][cb][r] = 0.f;

  const int stage_floats = (T.Kb + T.Nb) * 4 * WG_PIECE;
  const int nchunks = (T.tile_end - T.tile_begin) * 2;
  const unsigned smem_b = lds_byte_addr(smem);
  // Round 6 (as wgrad_bf16.hip): the CPW pieces a wave copies per chunk are the same (operand, block, quarter) for every chunk of
  // the task, so their wave-uniform source pointers live in SGPRs and walk the task -- + 4 KiB from a tile's first 32-row chunk to
  // its second, then on to the next tile -- instead of a 64-bit per-lane address computation per piece and chunk.
  const char* psrc[CPW];
  unsigned ymask = 0;      // bit i: piece i is a dY piece
  const int next_x = T.x_tile_stride * 4 - 4096, next_y = T.dy_tile_stride * 4 - 4096;   // bytes from a tile's second chunk to the next tile's first
  const int npx = T.Kb * 4, np = (T.Kb + T.Nb) * 4;
#pragma unroll
  for (int i = 0; i < CPW; ++i) {
    int p = wave + 8 * i;
    p = p < np ? p : np - 1;              // the tail re-copies the last piece: every wave issues CPW copies (one vmcnt count fits all)
    const bool isy = p >= npx;
    const int pp = isy ? p - npx : p;
    const int blk = pp >> 2, qq = pp & 3;
    const int stride = isy ? T.dy_tile_stride : T.x_tile_stride;
    const char* src = reinterpret_cast<const char*>((isy ? T.dY : T.X) + (size_t)T.tile_begin * stride) + (size_t)(blk * 8 + qq) * 1024;
    const unsigned long long u = reinterpret_cast<unsigned long long>(src);
    psrc[i] = reinterpret_cast<const char*>(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) |
                                            (unsigned)__builtin_amdgcn_readfirstlane((int)(u & 0xffffffffu)));
    ymask |= isy ? 1u << i : 0u;
  }
  ymask = (unsigned)__builtin_amdgcn_readfirstlane((int)ymask);
  int first = 1;                            // the next staged chunk is a tile's first
  unsigned stage_buf = smem_b;
  const unsigned ring_end = smem_b + (unsigned)(RING * stage_floats * 4);
  const unsigned voff = (unsigned)lane * 16u;
  auto stage_next = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
      int p = wave + 8 * i;
      p = p < np ? p : np - 1;
      lds_dma16s<WG_NT>(psrc[i], voff, stage_buf + (unsigned)p * (WG_PIECE * 4));
      psrc[i] += first ? 4096 : ((ymask >> i) & 1u) ? next_y : next_x;
    }
    first ^= 1;
    stage_buf += (unsigned)(stage_floats * 4);
    if (stage_buf >= ring_end) stage_buf = smem_b;
  };
  const bool active = kb0 < T.Kb;   // narrow K (Kb < number of k wave groups): surplus waves only stage

  for (int c = 0; c < RING - 1 && c < nchunks; ++c) stage_next();
  const float* Xs = smem;
  const float* const ring_floats_end = smem + RING * stage_floats;
  for (int ci = 0; ci < nchunks; ++ci) {
    if (ci + RING - 2 <= nchunks - 1) wait_vm<(RING - 2) * CPW>();   // chunk ci has landed, RING - 2 later ones may fly
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();     // ... for every wave, and nobody still reads the stage refilled next
    asm volatile("" ::: "memory");
    const bool refill = ci + RING - 1 < nchunks;
    if (active) {
      // waves w and w + 4 share a SIMD: one refills behind its first quarter of MFMAs, the other behind its second
      wgrad_compute<NRB>(acc, Xs, Xs + T.Kb * 4 * WG_PIECE, kb0, nb0, lane, wave >> 2,
                         [&]() __attribute__((always_inline)) { if (refill) stage_next(); });
    } else if (refill) {
      stage_next();
    }
    Xs += stage_floats;
    if (Xs >= ring_floats_end) Xs = smem;
  }
  if (!active) return;

  const int j = lane & 31, h = lane >> 5;
  const int ld = T.Nb * 32;
#pragma unroll
  for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int k = 32 * (kb0 + rb) + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        T.slab[(size_t)k * ld + 32 * (nb0 + cb) + j] = acc[rb][cb][reg];
      }
}

// Vector-column task (Nb == 0): vslab[kk][k][c] = sum_rows X[row][k] * vec[row][c] -- the weight
// gradients of the narrow heads (alpha: X = h8, vec.w ; rgb logits: X = rgb hidden, vec.xyz ;
// SE3 w and v heads: X = h6 read ONCE against vec = dL/dw and vec2 = dL/dv).  Thread (k = feature, kk): its float4s of a
// staged chunk are rows 4g..4g+3 with g = (qq&1) + 2 kk + 4 (qq>>1) of the chunk.  A stage = the X chunk (Kb x 4 KiB) + two
// 1 KiB vector pieces (32 rows x float4 in plain row order; lanes 32..63 of the copy fetch 512 B of slack never read).
template <int CPW, int RING>
__device__ __forceinline__ void wgrad_vec_body(const WgradTask& T, float* smem) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float va[4] = {0.f, 0.f, 0.f, 0.f}, vb[4] = {0.f, 0.f, 0.f, 0.f};
  const int nchunks = (T.tile_end - T.tile_begin) * 2;
  const int k = tid & 255, kk = tid >> 8;
  const int xfloats = T.Kb * 4 * WG_PIECE;
  const int stage_floats = xfloats + 2 * WG_PIECE;
  const bool two = T.vec2 != nullptr;
  const unsigned smem_b = lds_byte_addr(smem);
  auto stage = [&](int ci) {
    const int tile = T.tile_begin + (ci >> 1), c = ci & 1;
    const unsigned buf = smem_b + (unsigned)((ci % RING) * stage_floats * 4);
    stage_chunk<CPW>(T.X + (size_t)tile * T.x_tile_stride, T.Kb, nullptr, 0, c, buf, wave, lane);
    // waves 0 / 1 issue one copy more than CPW: their vmcnt(N) then waits for MORE than it has to (safe)
    const int r = lane & 31;
    if (wave == 0) lds_dma16<false>(T.vec + (size_t)tile * TILE_ROWS + 32 * c + r, buf + xfloats * 4);
    if (wave == 1 && two) lds_dma16<false>(T.vec2 + (size_t)tile * TILE_ROWS + 32 * c + r, buf + (xfloats + WG_PIECE) * 4);
  };
  for (int c = 0; c < RING - 1 && c < nchunks; ++c) stage(c);
  for (int ci = 0; ci < nchunks; ++ci) {
    if (ci + RING - 2 <= nchunks - 1) wait_vm<(RING - 2) * CPW>();
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (ci + RING - 1 < nchunks) stage(ci + RING - 1);
    if (k < T.Kb * 32) {
      const float* Xs = smem + (ci % RING) * stage_floats;
      const float* vs = Xs + xfloats;
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const float4 xv = *reinterpret_cast<const float4*>(Xs + ((((k >> 5) * 4 + qq) * 64) + (k & 31) + 32 * kk) * 4);
        const float xe[4] = {xv.x, xv.y, xv.z, xv.w};
        const int g = (qq & 1) + 2 * kk + 4 * (qq >> 1);   // granule within the 32-row chunk
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float4 dv = *reinterpret_cast<const float4*>(vs + 4 * (4 * g + e));
          va[0] = fmaf(xe[e], dv.x, va[0]); va[1] = fmaf(xe[e], dv.y, va[1]);
          va[2] = fmaf(xe[e], dv.z, va[2]); va[3] = fmaf(xe[e], dv.w, va[3]);
          if (two) {
            const float4 dw = *reinterpret_cast<const float4*>(vs + WG_PIECE + 4 * (4 * g + e));
            vb[0] = fmaf(xe[e], dw.x, vb[0]); vb[1] = fmaf(xe[e], dw.y, vb[1]);
            vb[2] = fmaf(xe[e], dw.z, vb[2]); vb[3] = fmaf(xe[e], dw.w, vb[3]);
          }
        }
      }
    }
  }
  if (k < T.Kb * 32) {
    *reinterpret_cast<float4*>(T.vslab + ((size_t)kk * T.Kb * 32 + k) * 4) = make_float4(va[0], va[1], va[2], va[3]);
    if (two) *reinterpret_cast<float4*>(T.vslab2 + ((size_t)kk * T.Kb * 32 + k) * 4) = make_float4(vb[0], vb[1], vb[2], vb[3]);
  }
}

__device__ __forceinline__ void wgrad_run_task(const WgradTask& T, float* smem) {
  if (T.Nb == 0) {
    if (T.Kb == 8)      wgrad_vec_body<4, 4>(T, smem);   // 34 KiB stages
    else if (T.Kb == 4) wgrad_vec_body<2, 8>(T, smem);   // 18 KiB
    else                wgrad_vec_body<4, 2>(T, smem);   // any Kb <= 8: two stages always fit
    return;
  }
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // 8 waves tile the [Kb][Nb] block grid: n-groups of 2 column blocks, the rest along k.
  const int ngn = T.Nb / 2;            // 2 or 4
  const int ngk = 8 / ngn;             // 4 or 2
  const int wn = wave % ngn, wk = wave / ngn;
  const int nrb = (T.Kb + ngk - 1) / ngk;   // 4, 2 or 1
  const int kb0 = wk * nrb, nb0 = 2 * wn;
  const int cpw = ((T.Kb + T.Nb) * 4 + 7) / 8;
  if (nrb == 4)      wgrad_body<4, 8, 2>(T, smem, kb0, nb0);   // 8 x 8 blocks: 64 KiB stages
  else if (nrb == 2) wgrad_body<2, 6, 3>(T, smem, kb0, nb0);   // 8 x 4: 48 KiB
  else if (cpw == 5) wgrad_body<1, 5, 4>(T, smem, kb0, nb0);   // 2 x 8 (1 x 8): 40 (36) KiB
  else if (cpw == 4) wgrad_body<1, 4, 5>(T, smem, kb0, nb0);   // 4 x 4: 32 KiB
  else if (cpw == 3) wgrad_body<1, 3, 6>(T, smem, kb0, nb0);   // 2 x 4 (1 x 4): 24 (20) KiB
  else               wgrad_body<1, 8, 2>(T, smem, kb0, nb0);   // anything else with one row block per wave
}

// One workgroup per CU; each walks its share of the linearised (layer, tile) work (equal cost per
// workgroup, so there is exactly one round and no tail), flushing a partial slab per segment.
__global__ __launch_bounds__(512) void wgrad_kernel(const WgradGroup* __restrict__ groups,
                                                    const WgradSegment* __restrict__ segs,
                                                    const int* __restrict__ seg_begin, float* __restrict__ ws,
                                                    unsigned long long* __restrict__ seg_clock) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int s0 = seg_begin[blockIdx.x], s1 = seg_begin[blockIdx.x + 1];
  for (int si = s0; si < s1; ++si) {
    const unsigned long long t0 = wall_clock64();
    const WgradSegment sg = segs[si];
    const WgradGroup G = groups[sg.group];
    WgradTask T;
    T.X = ws + G.x_off; T.x_kind = G.x_kind; T.x_tile_stride = G.x_tile_stride; T.x_kvalid = G.x_kvalid; T.Kb = G.Kb;
    T.dY = ws + G.dy_off; T.dy_kind = G.dy_kind; T.dy_tile_stride = G.dy_tile_stride; T.Nb = G.Nb;
    T.tile_begin = sg.tile_begin;
    T.tile_end = sg.tile_end;
    T.slab = ws + G.slab_off + (size_t)sg.slab_idx * (G.Kb * 32) * (G.Nb * 32);
    T.vec = G.vec_off >= 0 ? reinterpret_cast<const float4*>(ws + G.vec_off) : nullptr;
    T.vec2 = G.vec2_off >= 0 ? reinterpret_cast<const float4*>(ws + G.vec2_off) : nullptr;
    T.vslab = ws + G.vslab_off + (size_t)sg.slab_idx * 2 * (G.Kb * 32) * 4;
    T.vslab2 = ws + G.vslab2_off + (size_t)sg.slab_idx * 2 * (G.Kb * 32) * 4;
    wgrad_run_task(T, smem);
    __syncthreads();   // the next segment restages LDS
    if (seg_clock && threadIdx.x == 0) seg_clock[si] = wall_clock64() - t0;   // 100 MHz ticks (cost-model calibration)
  }
}

void launch_wgrad(const WgradGroup* d_groups, const WgradSegment* d_segs, const int* d_seg_begin, int nwg, float* ws,
                  unsigned long long* seg_clock, hipStream_t stream) {
  (void)hipFuncSetAttribute((const void*)wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WG_LDS_BYTES);
  hipLaunchKernelGGL(wgrad_kernel, dim3(nwg), dim3(512), WG_LDS_BYTES, stream, d_groups, d_segs, d_seg_begin, ws, seg_clock);
}

// dst[r][c] = sum_parts src[part][r][c].  Wide leaves (cols, ld multiples of 4) go 4 columns per
// thread with 4 independent partial sums so that the part loop keeps 4 x 16 B loads in flight.
// A workgroup column (blockIdx.y) runs a CHAIN of descriptors -- the passes that add into one destination, in pass order --: the
// element -> thread mapping (`path`) is the same along the chain, so a thread re-reads only what it wrote itself.
__global__ __launch_bounds__(256) void reduce_kernel(const ReduceDesc* __restrict__ descs, int first, const float* __restrict__ ws,
                                                     float* __restrict__ grad) {
  for (int di = first + (int)blockIdx.y; di >= 0;) {
    const ReduceDesc d = descs[di];
    di = d.next;
    if (d.path == 2) {
      // tall and thin (bias partials, one per chain-kernel workgroup): one wave per column, lanes stride over
      // the parts, shuffle tree at the end (deterministic order)
      const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
      for (int c = blockIdx.x * 4 + wv; c < d.cols; c += gridDim.x * 4) {
        const float* s = ws + d.src_off + c;
        float a0 = 0.f, a1 = 0.f;
        int q = lane;
        for (; q + 64 < d.nparts; q += 128) { a0 += s[(size_t)q * d.part_stride]; a1 += s[(size_t)(q + 64) * d.part_stride]; }
        if (q < d.nparts) a0 += s[(size_t)q * d.part_stride];
        float t = a0 + a1;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
        if (lane == 0) { float* o = grad + d.dst_off + c; *o = d.accumulate ? *o + t : t; }
      }
    } else if (d.path == 1) {
      const int c4 = d.cols >> 2, total = d.rows * c4;
      for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int r = idx / c4, c = (idx - r * c4) << 2;
        const float4* s = reinterpret_cast<const float4*>(ws + d.src_off + (size_t)r * d.src_ld + c);
        const size_t ps = (size_t)(d.part_stride >> 2);
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
        int q = 0;
        for (; q + 4 <= d.nparts; q += 4) {
          const float4 v0 = s[(q + 0) * ps], v1 = s[(q + 1) * ps], v2 = s[(q + 2) * ps], v3 = s[(q + 3) * ps];
          a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
          a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
          a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
          a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
        }
        for (; q < d.nparts; ++q) { const float4 v = s[q * ps]; a0.x += v.x; a0.y += v.y; a0.z += v.z; a0.w += v.w; }
        float4 t = make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z),
                               (a0.w + a1.w) + (a2.w + a3.w));
        float4* o = reinterpret_cast<float4*>(grad + d.dst_off + (size_t)r * d.dst_ld + c);
        if (d.accumulate) { const float4 p = *o; t.x += p.x; t.y += p.y; t.z += p.z; t.w += p.w; }
        *o = t;
      }
    } else {
      const int total = d.rows * d.cols;
      for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int r = idx / d.cols, c = idx - r * d.cols;
        const float* s = ws + d.src_off + (size_t)r * d.src_ld + c;
        float acc = 0.f;
        for (int q = 0; q < d.nparts; ++q) acc += s[(size_t)q * d.part_stride];
        float* o = grad + d.dst_off + (size_t)r * d.dst_ld + c;
        *o = d.accumulate ? *o + acc : acc;
      }
    }
  }
}

void launch_reduce(const ReduceDesc* d_table, int first, int ndesc, const float* ws, float* grad, hipStream_t stream) {
  hipLaunchKernelGGL(reduce_kernel, dim3(64, ndesc), dim3(256), 0, stream, d_table, first, ws, grad);
}

}  // namespace nrf
