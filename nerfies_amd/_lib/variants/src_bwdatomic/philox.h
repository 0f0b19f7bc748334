// Counter-based random numbers for the throughput runs (stand-in for the flax / threefry streams of the reference,
// models.py:275, 333, 355): Philox4x32-10 keyed by (seed, offset), one counter per (stream, element).
#pragma once
#include <stdint.h>

namespace nrf {

__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
  c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
}

__device__ __forceinline__ void philox4(uint64_t seed, uint64_t offset, uint32_t stream_id, uint32_t idx, uint32_t (&c)[4]) {
  c[0] = idx; c[1] = stream_id; c[2] = (uint32_t)offset; c[3] = (uint32_t)(offset >> 32);
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) { philox_round(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
}

// uniform in [0,1) for element `idx` of stream `stream_id`
__device__ __forceinline__ float philox_uniform(uint64_t seed, uint64_t offset, uint32_t stream_id, uint32_t idx) {
  uint32_t c[4];
  philox4(seed, offset, stream_id, idx, c);
  return (float)(c[0] >> 8) * (1.0f / 16777216.0f);
}

// standard normal (Box-Muller on two words of the same counter)
__device__ __forceinline__ float philox_normal(uint64_t seed, uint64_t offset, uint32_t stream_id, uint32_t idx) {
  uint32_t c[4];
  philox4(seed, offset, stream_id, idx, c);
  const float u1 = ((float)(c[0] >> 8) + 1.0f) * (1.0f / 16777216.0f);   // (0, 1]
  const float u2 = (float)(c[1] >> 8) * (1.0f / 16777216.0f);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

}  // namespace nrf
