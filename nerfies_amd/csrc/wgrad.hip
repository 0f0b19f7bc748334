// Weight-gradient GEMMs  dW[k][n] = sum_rows X[row][k] * dY[row][n]  for every dense layer
// of the NeRF MLPs (the transpose jax.grad builds for modules.MLP, modules.py:41-58).
//
// The reduction runs over rows (ray samples): 65k-196k per MLP, output only 256x256.  Each
// workgroup owns one split-K slice (a range of 128-row tiles) of one layer and produces a
// full [Kb*32][Nb*32] partial in registers: 8 waves x (<=4 x 2) 32x32 fp32 MFMA blocks, so X
// and dY stream from HBM exactly once.  Operands are staged through LDS in 32-row chunks
// (double buffered, one barrier per chunk); both are read feature-major so one ds_read_b128
// yields the operand of 4 MFMA k-steps.  Partials go to slabs, summed by reduce_kernel.
#include "nrf_internal.h"

namespace nrf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WG_LDP = 36;                    // LDS pitch (32 rows + 4 pad) -> conflict-free b128
constexpr int WG_OPER = 256 * WG_LDP;         // floats per operand per stage
constexpr int WG_STAGE = 2 * WG_OPER;
constexpr int WG_VEC = 2 * WG_STAGE;          // float offset of the [2 stages][32 rows] float4 vec staging

struct Granule { float4 v; int lds; };

// granule `gid` of 32-row chunk `c` of a tile: 4 consecutive rows of one feature.
__device__ __forceinline__ bool granule_src(const float* __restrict__ base, int kind, int kvalid, int nblocks,
                                            int c, int gid, const float4*& src, int& lds) {
  if (kind == SRC_FRAG256) {
    const int ln = gid & 63, q = (gid >> 6) & 3, blk = gid >> 8;   // blk = w*2+cb
    if (blk >= nblocks) return false;
    src = reinterpret_cast<const float4*>(base) + (blk * 16 + 4 * c + q) * 64 + ln;
    lds = (32 * blk + (ln & 31)) * WG_LDP + 4 * (q + 4 * (ln >> 5));
    return true;
  } else if (kind == SRC_FRAG128) {
    const int ln = gid & 63, q = (gid >> 6) & 3, blk = gid >> 8;   // blk = w
    if (blk >= nblocks) return false;
    src = reinterpret_cast<const float4*>(base) + (blk * 16 + 4 * c + q) * 64 + ln;
    lds = (32 * blk + (ln & 31)) * WG_LDP + 4 * (q + 4 * (ln >> 5));
    return true;
  } else {  // SRC_PLAIN: [k][128 rows]
    const int g = gid & 7, k = gid >> 3;
    if (k >= nblocks * 32) return false;
    lds = k * WG_LDP + 4 * g;
    src = (k < kvalid) ? reinterpret_cast<const float4*>(base + k * TILE_ROWS + 32 * c + 4 * g) : nullptr;
    return true;
  }
}

template <int NRB>
__device__ __forceinline__ void wgrad_compute(f32x16 (&acc)[NRB][2], const float* Xs, const float* Ys,
                                              int kb0, int nb0, int lane) {
  const int i = lane & 31, kk = lane >> 5;
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    float4 a[NRB], b[2];
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
      a[rb] = *reinterpret_cast<const float4*>(Xs + (32 * (kb0 + rb) + i) * WG_LDP + 8 * g4 + 4 * kk);
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
      b[cb] = *reinterpret_cast<const float4*>(Ys + (32 * (nb0 + cb) + i) * WG_LDP + 8 * g4 + 4 * kk);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) {
        const float av = s == 0 ? a[rb].x : s == 1 ? a[rb].y : s == 2 ? a[rb].z : a[rb].w;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          const float bv = s == 0 ? b[cb].x : s == 1 ? b[cb].y : s == 2 ? b[cb].z : b[cb].w;
          acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[rb][cb], 0, 0, 0);
        }
      }
    }
  }
}

// narrow dY columns on the VALU: va[c] += sum_rows X[row][k] * vec[row][c] for this thread's k and 16 rows
__device__ __forceinline__ void vec_accumulate(float (&va)[4], const float* Xs, const float* vs, int kmax) {
  const int k = threadIdx.x & 255, hf = threadIdx.x >> 8;
  if (k >= kmax) return;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 xv = *reinterpret_cast<const float4*>(Xs + k * WG_LDP + 16 * hf + 4 * g);
    const float xe[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float4 dv = *reinterpret_cast<const float4*>(vs + 4 * (16 * hf + 4 * g + e));
      va[0] = fmaf(xe[e], dv.x, va[0]); va[1] = fmaf(xe[e], dv.y, va[1]);
      va[2] = fmaf(xe[e], dv.z, va[2]); va[3] = fmaf(xe[e], dv.w, va[3]);
    }
  }
}

template <int NRB>
__device__ __forceinline__ void wgrad_body(const WgradTask& T, float* smem, int kb0, int nb0) {
  const int tid = threadIdx.x, lane = tid & 63;
  f32x16 acc[NRB][2];
#pragma unroll
  for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][cb][r] = 0.f;

  const int nchunks = (T.tile_end - T.tile_begin) * 4;
  float4 rx[4], ry[4];
  int lx[4], ly[4];
  bool vx[4], vy[4];

  auto fetch = [&](int ci) {
    const int tile = T.tile_begin + (ci >> 2), c = ci & 3;
    const float* xb = T.X + (size_t)tile * T.x_tile_stride;
    const float* yb = T.dY + (size_t)tile * T.dy_tile_stride;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const float4* s;
      vx[m] = granule_src(xb, T.x_kind, T.x_kvalid, T.Kb, c, tid + 512 * m, s, lx[m]);
      rx[m] = (vx[m] && s) ? *s : make_float4(0.f, 0.f, 0.f, 0.f);
      vy[m] = granule_src(yb, T.dy_kind, 1 << 30, T.Nb, c, tid + 512 * m, s, ly[m]);
      ry[m] = (vy[m] && s) ? *s : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto commit = [&](int stage) {
    float* Xs = smem + stage * WG_STAGE;
    float* Ys = Xs + WG_OPER;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (vx[m]) *reinterpret_cast<float4*>(Xs + lx[m]) = rx[m];
      if (vy[m]) *reinterpret_cast<float4*>(Ys + ly[m]) = ry[m];
    }
  };

  if (nchunks > 0) { fetch(0); commit(0); }
  __syncthreads();
  const bool active = kb0 < T.Kb;   // narrow K (Kb < number of k wave groups): surplus waves only stage
  for (int ci = 0; ci < nchunks; ++ci) {
    const bool more = ci + 1 < nchunks;
    if (more) fetch(ci + 1);
    const float* Xs = smem + (ci & 1) * WG_STAGE;
    if (active) wgrad_compute<NRB>(acc, Xs, Xs + WG_OPER, kb0, nb0, lane);
    if (more) commit((ci + 1) & 1);
    __syncthreads();
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

// Vector-column task (Nb == 0): vslab[hf][k][c] = sum_rows X[row][k] * vec[row][c] -- the weight
// gradients of the two narrow heads (alpha: X = h8, vec.w ; rgb logits: X = rgb hidden, vec.xyz).
__device__ __forceinline__ void wgrad_vec_body(const WgradTask& T, float* smem) {
  const int tid = threadIdx.x;
  float va[4] = {0.f, 0.f, 0.f, 0.f};
  const int nchunks = (T.tile_end - T.tile_begin) * 4;
  float4 rx[4], rvec = make_float4(0.f, 0.f, 0.f, 0.f);
  int lx[4];
  bool vx[4];
  auto fetch = [&](int ci) {
    const int tile = T.tile_begin + (ci >> 2), c = ci & 3;
    const float* xb = T.X + (size_t)tile * T.x_tile_stride;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const float4* s;
      vx[m] = granule_src(xb, T.x_kind, T.x_kvalid, T.Kb, c, tid + 512 * m, s, lx[m]);
      rx[m] = (vx[m] && s) ? *s : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (tid < 32) rvec = T.vec[(size_t)tile * TILE_ROWS + 32 * c + tid];
  };
  auto commit = [&](int stage) {
    float* Xs = smem + stage * WG_STAGE;
#pragma unroll
    for (int m = 0; m < 4; ++m)
      if (vx[m]) *reinterpret_cast<float4*>(Xs + lx[m]) = rx[m];
    if (tid < 32) *reinterpret_cast<float4*>(smem + WG_VEC + stage * 128 + 4 * tid) = rvec;
  };
  if (nchunks > 0) { fetch(0); commit(0); }
  __syncthreads();
  for (int ci = 0; ci < nchunks; ++ci) {
    const bool more = ci + 1 < nchunks;
    if (more) fetch(ci + 1);
    vec_accumulate(va, smem + (ci & 1) * WG_STAGE, smem + WG_VEC + (ci & 1) * 128, T.Kb * 32);
    if (more) commit((ci + 1) & 1);
    __syncthreads();
  }
  const int k = tid & 255, hf = tid >> 8;
  if (k < T.Kb * 32)
    *reinterpret_cast<float4*>(T.vslab + ((size_t)hf * T.Kb * 32 + k) * 4) = make_float4(va[0], va[1], va[2], va[3]);
}

__device__ __forceinline__ void wgrad_run_task(const WgradTask& T, float* smem) {
  if (T.Nb == 0) { wgrad_vec_body(T, smem); return; }
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // 8 waves tile the [Kb][Nb] block grid: n-groups of 2 column blocks, the rest along k.
  const int ngn = T.Nb / 2;            // 2 or 4
  const int ngk = 8 / ngn;             // 4 or 2
  const int wn = wave % ngn, wk = wave / ngn;
  const int nrb = (T.Kb + ngk - 1) / ngk;   // 4, 2 or 1
  const int kb0 = wk * nrb, nb0 = 2 * wn;
  if (nrb == 4)      wgrad_body<4>(T, smem, kb0, nb0);
  else if (nrb == 2) wgrad_body<2>(T, smem, kb0, nb0);
  else               wgrad_body<1>(T, smem, kb0, nb0);
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
    T.vslab = ws + G.vslab_off + (size_t)sg.slab_idx * 2 * (G.Kb * 32) * 4;
    wgrad_run_task(T, smem);
    __syncthreads();   // the next segment restages LDS
    if (seg_clock && threadIdx.x == 0) seg_clock[si] = wall_clock64() - t0;   // 100 MHz ticks (cost-model calibration)
  }
}

void launch_wgrad(const WgradGroup* d_groups, const WgradSegment* d_segs, const int* d_seg_begin, int nwg, float* ws,
                  unsigned long long* seg_clock, hipStream_t stream) {
  const size_t lds = (size_t)(2 * WG_STAGE + 256) * sizeof(float);
  (void)hipFuncSetAttribute((const void*)wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(wgrad_kernel, dim3(nwg), dim3(512), lds, stream, d_groups, d_segs, d_seg_begin, ws, seg_clock);
}

// dst[r][c] = sum_parts src[part][r][c]
__global__ void reduce_kernel(const ReduceDesc* __restrict__ descs, const float* __restrict__ ws,
                              float* __restrict__ grad) {
  const ReduceDesc d = descs[blockIdx.y];
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

void launch_reduce(const ReduceDesc* d_descs, int ndesc, const float* ws, float* grad, hipStream_t stream) {
  hipLaunchKernelGGL(reduce_kernel, dim3(32, ndesc), dim3(256), 0, stream, d_descs, ws, grad);
}

}  // namespace nrf
