// Register-resident forward chain of the NeRF MLP (inference / rendering path, no activation stash).
//
// Same function as nerf_mlp_fwd_kernel<false> (mlp_chain.hip; reference modules.py:26-169, models.py:270-277), other
// dataflow: the GEMMs are transposed,  H^T[out feature][sample] = W^T . X^T,  and ONE wave owns 32 samples and ALL
// output features of every layer.  In v_mfma_f32_32x32x2_f32's D layout lane (n, h) (n = lane % 32, h = lane / 32)
// holds, in accumulator register 4j+i of output block o, feature 32o + 8j + 4h + i of sample n -- and the B operand
// of the next layer's k-step wants, in lane (n, h), the value X[n][k] for k-slot h.  So with the K order
//     k-step t = 16b + 4j + i,  k-slot h   <->   feature 32b + 8j + 4h + i
// accumulator register (b, 4j+i) IS the B operand of k-step t after the ReLU: activations never leave the register
// file, there is no LDS traffic and no barrier; the weights (A operand, packed in that K order by rc_pack_kernel)
// stream from L2 at the same bytes per MFMA-clock per CU as in the LDS-tiled kernel.  One wave per SIMD (128 input +
// 128 accumulator registers + weights), four independent waves per workgroup.
//
// The alpha head rides as a ninth output block of the bottleneck GEMM (same input), the rgb logits as a one-block GEMM.
#include "chain_common.h"

namespace nrf {

namespace {

// The weight stream: every GEMM of the chain reads its A operands from ONE buffer, laid out in execution order in
// groups of NOUT float4 per lane (1 KiB per output block): [bias group] + 4 NIN groups per GEMM.  A group is fetched
// while the previous one is being multiplied, across GEMM boundaries too (the last group of a GEMM prefetches the
// first group of the next one), so a wave never starts a layer on a cold load.  Buffer loads with one per-lane
// voffset and a wave-uniform scalar offset keep the address arithmetic off the VALU and out of the register file.
// (kept as separate scalars / a plain register array: wrapped in a struct hipcc demotes the weights to LDS and
// the descriptor to VGPRs, which turns every load into a readfirstlane waterfall loop.)
__device__ __forceinline__ float4 rc_ld(__amdgpu_buffer_rsrc_t rs, int voff, int soff, int o) {
  const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff + o * 1024, 0);
  return make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
}

// acc[o] (+)= W^T-panel x in.  w[]: the group about to be consumed (in/out); soff: its byte offset in the stream.
// BIAS: the GEMM starts with a bias group (A = bias[m] in k-slot 0, B = 1) on a zero accumulator, so neither a bias
// load nor an accumulator init is needed.  NNEXT: width of the group that follows this GEMM in the stream; WRAP: the
// stream restarts (last GEMM of the chain).
template <int NIN, int NOUT, int NNEXT, bool BIAS, bool WRAP = false>
__device__ __forceinline__ void rc_gemm(f32x16 (&acc)[NOUT], const float (&in)[NIN][16], __amdgpu_buffer_rsrc_t rs, int voff,
                                        int& soff, float4 (&w)[9]) {
  if (BIAS) {
    soff += NOUT * 1024;
    float4 wn[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) wn[o] = rc_ld(rs, voff, soff, o);
    __builtin_amdgcn_sched_barrier(0);
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int o = 0; o < NOUT; ++o) acc[o] = mfma32(w[o].x, 1.0f, zero);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int o = 0; o < NOUT; ++o) w[o] = wn[o];
  }
#pragma unroll
  for (int tg = 0; tg < NIN * 4; ++tg) {
    constexpr int NG = NIN * 4;
    const int cnt = tg == NG - 1 ? NNEXT : NOUT;
    soff = (tg == NG - 1 && WRAP) ? 0 : soff + NOUT * 1024;
    float4 wn[9];
#pragma unroll
    for (int o = 0; o < 9; ++o)
      if (o < cnt) wn[o] = rc_ld(rs, voff, soff, o);
    __builtin_amdgcn_sched_barrier(0);   // the next group is in flight before this group's MFMAs start
    const int b = tg >> 2, j = tg & 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float x = in[b][4 * j + i];
#pragma unroll
      for (int o = 0; o < NOUT; ++o) {
        const float wv = i == 0 ? w[o].x : i == 1 ? w[o].y : i == 2 ? w[o].z : w[o].w;
        acc[o] = mfma32(wv, x, acc[o]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int o = 0; o < 9; ++o)
      if (o < cnt) w[o] = wn[o];
  }
}

// one v_med3_f32 (fmaxf costs a canonicalising v_max on top)
__device__ __forceinline__ float rc_relu(float x) { return __builtin_amdgcn_fmed3f(x, 0.f, __builtin_inff()); }

// condterm[feature] for this lane's 16 features of block o: 4 float4s (features 32o + 8j + 4h .. +3)
__device__ __forceinline__ void rc_bias(f32x16& acc, const float* __restrict__ bias, int o, int h) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 b = *reinterpret_cast<const float4*>(bias + 32 * o + 8 * j + 4 * h);
    acc[4 * j] = b.x; acc[4 * j + 1] = b.y; acc[4 * j + 2] = b.z; acc[4 * j + 3] = b.w;
  }
}

__device__ __forceinline__ float rc_sigma(float x, int kind) {
  return kind == 1 ? fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))) : fmaxf(x, 0.f);
}

}  // namespace

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void nerf_mlp_fwd_reg_kernel(const ChainFwdArgs A) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 31, h = lane >> 5;
  const float* __restrict__ prm = A.params;
  const int ngroups = (A.rows + 31) / 32;   // 32-sample groups, one per wave iteration
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(A.wpk, A.rc.total * 4);
  const int voff = lane * 16;
  int soff = 0;
  float4 w[9];
#pragma unroll
  for (int o = 0; o < 8; ++o) w[o] = rc_ld(rs, voff, soff, o);   // bias group of layer 0

#pragma unroll 1
  for (int grp = blockIdx.x * 4 + wave; grp < ngroups; grp += gridDim.x * 4) {
    const int row = grp * 32 + n;
    const int r = row < A.rows ? row : A.rows - 1;
    // ---- sample point + SinusoidalEncoder (modules.py:213-228), straight into B-operand registers ----
    float x[3];
    if (A.points) {
      x[0] = A.points[3 * r]; x[1] = A.points[3 * r + 1]; x[2] = A.points[3 * r + 2];
    } else {
      const int ray = r / A.S;
      const float z = A.zvals[r];
#pragma unroll
      for (int c = 0; c < 3; ++c) x[c] = __fadd_rn(A.origins[3 * ray + c], __fmul_rn(z, A.directions[3 * ray + c]));
    }
    const float half_pi = 1.57079632679489661923f;
    // formed twice (layer 0 and the skip layer) rather than kept in 32 registers across four layers
    auto posenc = [&](float (&pe)[2][16]) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int e = 32 * b + 8 * (q >> 2) + 4 * h + (q & 3);   // posenc feature held in this register
          float v = 0.f;
          if (e < 3) {
            v = e == 0 ? x[0] : e == 1 ? x[1] : x[2];
          } else if (e < A.P) {
            const int idx = e - 3, f = idx / 6, rem = idx - 6 * f, c = rem >= 3 ? rem - 3 : rem;
            const float a = __fmul_rn(c == 0 ? x[0] : c == 1 ? x[1] : x[2], (float)(1 << f));
            v = sinf(rem >= 3 ? __fadd_rn(a, half_pi) : a);
          }
          pe[b][q] = v;
        }
    };

    // ---- trunk: 8 x Dense(256)+ReLU, skip concat [h, posenc] at layer 4 (modules.py:41-50) ----
    float act[8][16];
    f32x16 acc[8];
    {
      float pe[2][16];
      posenc(pe);
      rc_gemm<2, 8, 8, true>(acc, pe, rs, voff, soff, w);
#pragma unroll
      for (int o = 0; o < 8; ++o)
#pragma unroll
        for (int q = 0; q < 16; ++q) act[o][q] = rc_relu(acc[o][q]);
    }
#pragma unroll 1
    for (int l = 1; l < TRUNK_DEPTH; ++l) {
      rc_gemm<8, 8, 9, true>(acc, act, rs, voff, soff, w);   // followed by 8 (a trunk layer / the skip rows) or 9 (bottleneck) blocks
      if (l == SKIP_LAYER) {
        float pe[2][16];
        posenc(pe);
        rc_gemm<2, 8, 8, false>(acc, pe, rs, voff, soff, w);
      }
#pragma unroll
      for (int o = 0; o < 8; ++o)
#pragma unroll
        for (int q = 0; q < 16; ++q) act[o][q] = rc_relu(acc[o][q]);
    }

    // ---- bottleneck Dense(256) (no activation) + alpha head Dense(256->1) as a ninth output block ----
    float bn[8][16];
    float alpha_raw;
    {
      f32x16 acc9[9];
      rc_gemm<8, 9, 4, true>(acc9, act, rs, voff, soff, w);
#pragma unroll
      for (int o = 0; o < 8; ++o)
#pragma unroll
        for (int q = 0; q < 16; ++q) bn[o][q] = acc9[o][q];
      alpha_raw = acc9[8][0];   // valid in lanes h == 0 (feature 0 of the block); its bias rode in the bias group
    }

    // ---- rgb branch: Dense(256+R -> 128)+ReLU (per-ray condition term precomputed), Dense(128 -> 3) ----
    float rgbh[4][16];
    {
      f32x16 acc4[4];
      const int ray = min(r / A.S, A.B - 1);
      const float* ct = A.condterm + (size_t)ray * RGB_W;
#pragma unroll
      for (int o = 0; o < 4; ++o) rc_bias(acc4[o], ct, o, h);   // condterm already contains the bias
      rc_gemm<8, 4, 1, false>(acc4, bn, rs, voff, soff, w);
#pragma unroll
      for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int q = 0; q < 16; ++q) rgbh[o][q] = rc_relu(acc4[o][q]);
    }
    f32x16 acc1[1];
    rc_gemm<4, 1, 8, true, true>(acc1, rgbh, rs, voff, soff, w);
    if (h == 0 && row < A.rows) {   // features 0..2 of the logit block and the alpha feature live in lanes h == 0
      float4 o;
      o.x = 1.f / (1.f + expf(-acc1[0][0]));
      o.y = 1.f / (1.f + expf(-acc1[0][1]));
      o.z = 1.f / (1.f + expf(-acc1[0][2]));
      o.w = rc_sigma(alpha_raw, A.sigma_act);
      A.out4[row] = o;
    }
  }
}

namespace {
// One descriptor fills `ngroups` consecutive groups of the stream, columns o0 .. o0 + nout of a stream that is
// nout_panel blocks wide.  kind 0 (weights): group g = 4b + j, float4 component i = W[row0 + k][32 o + m] with
// k = 32b + 8j + 4h + i (m = lane % 32, h = lane / 32), zero outside [0, krows) x [0, ncols).  kind 1 (bias group):
// .x = bias[32 o + m] in the lanes of k-slot 0, everything else zero.
__global__ __launch_bounds__(256) void rc_pack_kernel(const RcPackDesc* __restrict__ descs, const float* __restrict__ params,
                                                      float* __restrict__ ws) {
  const RcPackDesc d = descs[blockIdx.y];
  const int total = d.ngroups * d.nout * 64;
  float4* dst = reinterpret_cast<float4*>(ws + d.dst_off);
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int lane = idx & 63, o = (idx >> 6) % d.nout, g = (idx >> 6) / d.nout;
    const int m = lane & 31, h = lane >> 5, b = g >> 2, j = g & 3;
    const int col = 32 * o + m;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (d.kind == 1) {
      if (h == 0 && col < d.ncols) v[0] = params[d.src_off + col];
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = 32 * b + 8 * j + 4 * h + i;
        if (k < d.krows && col < d.ncols) v[i] = params[d.src_off + (int64_t)(d.row0 + k) * d.src_ld + col];
      }
    }
    dst[(size_t)(g * d.nout_panel + d.o0 + o) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
  }
}
}  // namespace

void launch_rc_pack(const RcPackDesc* descs, int ndesc, const float* params, float* ws, hipStream_t stream) {
  if (ndesc > 0) rc_pack_kernel<<<dim3(32, ndesc), 256, 0, stream>>>(descs, params, ws);
}

void launch_chain_fwd_reg(const ChainFwdArgs& a, int grid, hipStream_t stream) {
  hipLaunchKernelGGL(nerf_mlp_fwd_reg_kernel, dim3(grid), dim3(256), 0, stream, a);
}

}  // namespace nrf
