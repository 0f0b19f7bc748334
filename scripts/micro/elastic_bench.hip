// elastic_kernel (csrc/warp_chain.hip) alone, at the config-D shape (512 rays x 256 coarse samples = 131072 rows) on synthetic
// head outputs: what the kernel costs without the step around it, and which part of it.  Build (one binary per variant):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Inerfies_amd/csrc -Iinclude [-DNRF_EL_...] scripts/micro/elastic_bench.hip \
//         nerfies_amd/csrc/warp_chain.hip -o scripts/micro/_bin/elastic_bench[_variant]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "nrf_internal.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int rows = argc > 1 ? atoi(argv[1]) : 131072;
  const int rows_pad = (rows + 255) / 256 * 256;
  const int loss_type = argc > 2 ? atoi(argv[2]) : NRF_ELASTIC_LOG_SVALS;
  std::vector<float> x(3 * (size_t)rows), wv(8 * (size_t)rows), tan(3 * 8 * (size_t)rows_pad), coef(rows);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.f - 0.5f; };
  for (auto& v : x) v = rnd();
  for (auto& v : wv) v = 0.6f * rnd();        // |w| ~ 0.3: the closed-form branch
  for (auto& v : tan) v = 0.8f * rnd();
  for (auto& v : coef) v = 0.5f + rnd();
  float *dx, *dwv, *dtan, *dcoef, *ddw, *ddv, *dpw, *dpv, *dsums;
  CK(hipMalloc(&dx, x.size() * 4)); CK(hipMalloc(&dwv, wv.size() * 4)); CK(hipMalloc(&dtan, tan.size() * 4)); CK(hipMalloc(&dcoef, coef.size() * 4));
  CK(hipMalloc(&ddw, 3 * 16 * (size_t)rows_pad)); CK(hipMalloc(&ddv, 3 * 16 * (size_t)rows_pad));
  CK(hipMalloc(&dpw, 16 * (size_t)rows_pad)); CK(hipMalloc(&dpv, 16 * (size_t)rows_pad)); CK(hipMalloc(&dsums, 5 * 4 * (size_t)(rows_pad / 256 + 1)));
  CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dwv, wv.data(), wv.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dtan, tan.data(), tan.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dcoef, coef.data(), coef.size() * 4, hipMemcpyHostToDevice));
  nrf::ElasticArgs a{};
  a.x_rows = dx; a.prim_win = nullptr; a.prim_wv = (const float4*)dwv; a.tan_wv = (const float4*)dtan; a.coef = dcoef;
  a.tan_dw4 = (float4*)ddw; a.tan_dv4 = (float4*)ddv; a.prim_dw4 = (float4*)dpw; a.prim_dv4 = (float4*)dpv; a.part = dsums;
  a.rows = rows; a.rows_pad = rows_pad; a.PKS = 0; a.eps = 1e-6f; a.alpha = -2.f; a.scale = 0.03f; a.gscale = 1e-3f / 512; a.inv_rays = 1.f / 512;
  a.dyn = nullptr; a.res_selected = 0; a.loss_type = loss_type;
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) nrf::launch_elastic(a, st);
  CK(hipStreamSynchronize(st));
  const int N = 50;
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < N; ++i) nrf::launch_elastic(a, st);
  CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const int nwg = (rows_pad + 255) / 256;
  std::vector<float> part(5 * (size_t)nwg);
  CK(hipMemcpy(part.data(), dsums, part.size() * 4, hipMemcpyDeviceToHost));
  double sums[5] = {0, 0, 0, 0, 0};
  for (int q = 0; q < 5; ++q) for (int w = 0; w < nwg; ++w) sums[q] += part[(size_t)q * nwg + w];
  std::vector<float> pw(4 * (size_t)rows_pad), tdw(4 * (size_t)rows_pad);
  CK(hipMemcpy(pw.data(), dpw, pw.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(tdw.data(), ddw, tdw.size() * 4, hipMemcpyDeviceToHost));
  double c1 = 0, c2 = 0;
  for (size_t i = 0; i < pw.size(); ++i) { c1 += pw[i]; c2 += tdw[i]; }
  printf("elastic_kernel rows %d type %d: %.2f us per launch   sums %.6e %.6e %.6e  checksums %.9e %.9e\n", rows, loss_type, 1e3 * ms / N,
         sums[0], sums[1], sums[2], c1, c2);
  return 0;
}
