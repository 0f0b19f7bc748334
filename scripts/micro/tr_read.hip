// Semantics check of ds_read_b64_tr_b16 (gfx950) as wgrad_bf16.hip uses it: every lane passes the address of its own
// 8-byte piece (4 bf16); within each group of 16 lanes the 16x4 block is transposed:
//   result[lane c of the group][j] = piece[4j + (c >> 2)][c & 3].
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 scripts/micro/tr_read.hip -o /tmp/tr_read && /tmp/tr_read
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned short* in, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[256];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = in[i];
  __syncthreads();
  const int l = threadIdx.x;
  v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(lds + 4 * l));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = r[j];
}
int main() {
  unsigned short h[256], o[256], *di, *dout;
  for (int i = 0; i < 256; ++i) h[i] = i;
  hipMalloc(&di, 512); hipMalloc(&dout, 512);
  hipMemcpy(di, h, 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
  hipMemcpy(o, dout, 512, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 4; ++j) {
      const int c = l & 15, piece = 16 * (l >> 4) + 4 * j + (c >> 2);
      if (o[l * 4 + j] != 4 * piece + (c & 3)) ++bad;
    }
  printf("tr_read: %d mismatches vs result[c][j] = piece[4j + (c>>2)][c&3]\n", bad);
  if (bad) for (int l = 0; l < 20; ++l) printf("lane %2d: %3d %3d %3d %3d\n", l, o[4 * l], o[4 * l + 1], o[4 * l + 2], o[4 * l + 3]);
  return bad != 0;
}
