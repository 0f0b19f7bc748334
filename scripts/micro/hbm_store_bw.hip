// What HBM3E on one MI355X sustains for the access patterns of the bf16 training kernels' activation stash: each wave issues
// 1 KiB buffer stores (16 B per lane, non-temporal, consecutive KiB of a private region) -- exactly bf_store16 of csrc/bf16_chain.h
// -- from 256 workgroups x 8 waves, one workgroup per CU.  Variants: store only (the forward / reverse chains), load only
// (wgrad_bf16's side), and 1:1 load + store (a copy).  Each variant moves 8 GiB so that neither the 32 MiB of L2 nor the 256 MiB
// Infinity Cache can absorb it.  Round 6: the SE3 training kernels write 3.4-4.7 TB/s (rocprofv3 WRITE_SIZE / duration); this says
// how much of the ceiling that is.
// Build + run:  hipcc --offload-arch=gfx950 -O3 scripts/micro/hbm_store_bw.hip -o gpurun_out/hbm_store_bw && gpurun_out/hbm_store_bw
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(p, 0, bytes, 0x00020000);
}

// MODE 0: store (aux = NT), 1: store (plain), 2: load only, 3: copy (load + nt store)
// every wave owns a contiguous slice of `per_wave` bytes and walks it in 1-KiB wave instructions, UNR in flight
template <int MODE>
__global__ __launch_bounds__(512) void stream(char* dst, const char* src, size_t per_wave, unsigned* sink) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * 8 + (threadIdx.x >> 6);
  char* d = dst + wave * per_wave;
  const char* s = src + wave * per_wave;
  u32x4 v = {(unsigned)lane, 0x3F803F80u, (unsigned)wave, 7u};
  u32x4 acc = {0, 0, 0, 0};
  constexpr int UNR = 8;
  for (size_t off = 0; off < per_wave; off += UNR * 1024) {
    // the stash kernels re-make their descriptor per 4 KiB panel; here one per 8 KiB
    const __amdgpu_buffer_rsrc_t rd = rsrc(d + off, UNR * 1024);
    const __amdgpu_buffer_rsrc_t rs = rsrc(const_cast<char*>(s) + off, UNR * 1024);
    u32x4 t[UNR];
    if (MODE >= 2) {
#pragma unroll
      for (int u = 0; u < UNR; ++u) t[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16 + u * 1024, 0, MODE == 2 ? 2 : 0);
    }
    if (MODE == 2) {
#pragma unroll
      for (int u = 0; u < UNR; ++u) acc ^= t[u];
    } else {
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const u32x4 q = MODE == 3 ? t[u] : v;
        __builtin_amdgcn_raw_buffer_store_b128(q, rd, lane * 16 + u * 1024, 0, MODE == 1 ? 0 : 2);
      }
      v.x += 64;
    }
  }
  if (MODE == 2 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <int MODE>
void run(const char* what, char* a, char* b, size_t bytes, unsigned* sink) {
  const int grid = 256;
  const size_t per_wave = bytes / (grid * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {   // second repetition is the quoted one (first touches the pages)
    hipEventRecord(e0);
    for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(stream<MODE>, dim3(grid), dim3(512), 0, 0, a, b, per_wave, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double moved = 4.0 * bytes * (MODE == 3 ? 2 : 1);
    if (rep) printf("%-44s %7.3f ms per %5.2f GiB launch  %7.1f GB/s%s\n", what, ms / 4, bytes / 1073741824.0, moved / (ms * 1e-3) / 1e9,
                    MODE == 3 ? " (read + written)" : "");
  }
}

int main() {
  const size_t bytes = (size_t)2 << 30;   // per launch; four launches per measurement
  char *a, *b;
  unsigned* sink;
  if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
  hipDeviceSynchronize();
  run<0>("store only, 1 KiB nt buffer stores", a, b, bytes, sink);
  run<1>("store only, plain buffer stores", a, b, bytes, sink);
  run<2>("load only, 1 KiB nt buffer loads", a, b, bytes, sink);
  run<3>("copy: load + nt store", a, b, bytes, sink);
  // the same with a working set the Infinity Cache holds (128 MiB): what a stash slice that stays on die would see
  run<0>("store only, nt, 128 MiB region (4 x over)", a, b, (size_t)128 << 20, sink);
  run<2>("load only, nt, 128 MiB region (4 x over)", a, b, (size_t)128 << 20, sink);
  return 0;
}
